"""Tokenization and chat templating.

The reference loads `AutoTokenizer.from_pretrained(MODEL_NAME)` (pad <- eos fallback,
orchestration.py:34-36), wraps the user message in the TinyLlama/Zephyr chat template with a
fixed system prompt (orchestration.py:60-67) and decodes with `skip_special_tokens=True`
(orchestration.py:199). Same here, plus a byte-level fallback tokenizer for runs with no
tokenizer files (random-init benchmarks; the Hugging Face hub is not reachable).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

ZEPHYR_TEMPLATE = "<|system|>\n{system}</s>\n<|user|>\n{user}</s>\n<|assistant|>\n"


def format_chat_prompt(user_message: str, system_prompt: str = "You are a helpful assistant.") -> str:
    """The reference's prompt format, byte for byte (orchestration.py:66)."""
    return ZEPHYR_TEMPLATE.format(system=system_prompt, user=user_message)


class ByteTokenizer:
    """UTF-8 bytes as tokens. ids: 0 pad, 1 bos, 2 eos, 3.. = byte value + 3.
    The literal strings "<s>" and "</s>" map to bos / eos so chat templates keep their meaning."""

    pad_token_id, bos_token_id, eos_token_id = 0, 1, 2
    OFFSET = 3
    name = "byte-fallback"

    def __init__(self, vocab_size: int = 259, eos_token_id: Optional[int] = None, bos_token_id: Optional[int] = None):
        if vocab_size < 259:
            raise ValueError("byte tokenizer needs a vocabulary of at least 259 entries")
        self.vocab_size = vocab_size
        # honour the model's own special ids when they do not collide with byte ids' meaning
        if eos_token_id is not None and 0 <= eos_token_id < vocab_size:
            self.eos_token_id = eos_token_id
        if bos_token_id is not None and 0 <= bos_token_id < vocab_size:
            self.bos_token_id = bos_token_id
        self.chat_template = None
        self.bos_token = "<s>"

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        ids: List[int] = [self.bos_token_id] if add_special_tokens else []
        i = 0
        while i < len(text):
            if text.startswith("</s>", i):
                ids.append(self.eos_token_id); i += 4
            elif text.startswith("<s>", i):
                ids.append(self.bos_token_id); i += 3
            else:
                ids += [b + self.OFFSET for b in text[i].encode("utf-8")]
                i += 1
        return ids

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        out = bytearray()
        specials = {self.pad_token_id: "", self.bos_token_id: "<s>", self.eos_token_id: "</s>"}
        for t in ids:
            t = int(t)
            if t in specials:
                if not skip_special_tokens:
                    out += specials[t].encode()
            elif self.OFFSET <= t < self.OFFSET + 256:
                out.append(t - self.OFFSET)
            # ids beyond the byte range (random-init models sample them) have no text
        return out.decode("utf-8", errors="replace")


class HFTokenizer:
    def __init__(self, path: str):
        from transformers import AutoTokenizer
        self.tk = AutoTokenizer.from_pretrained(path, local_files_only=True)
        if self.tk.pad_token is None:                      # orchestration.py:35-36
            self.tk.pad_token = self.tk.eos_token
        self.name = path
        self.eos_token_id = self.tk.eos_token_id
        self.bos_token_id = self.tk.bos_token_id
        self.pad_token_id = self.tk.pad_token_id
        self.vocab_size = len(self.tk)
        self.chat_template = getattr(self.tk, "chat_template", None)
        self.bos_token = self.tk.bos_token

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        return list(self.tk.encode(text, add_special_tokens=add_special_tokens))

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        return self.tk.decode(list(ids), skip_special_tokens=skip_special_tokens)

    def apply_chat_template(self, user_message: str, system_prompt: str) -> str:
        msgs = [{"role": "system", "content": system_prompt}, {"role": "user", "content": user_message}]
        return self.tk.apply_chat_template(msgs, tokenize=False, add_generation_prompt=True)


_TOKENIZER_FILES = ("tokenizer.json", "tokenizer.model", "vocab.json", "tokenizer_config.json")


def load_tokenizer(path: Optional[str], vocab_size: int, eos_token_id: Optional[int] = None,
                   bos_token_id: Optional[int] = None):
    if path and os.path.isdir(path) and any(os.path.exists(os.path.join(path, f)) for f in _TOKENIZER_FILES):
        return HFTokenizer(path)
    # keep the byte range intact: model-declared specials that fall inside it are ignored
    eos = eos_token_id if eos_token_id is not None and (eos_token_id < 3 or eos_token_id >= 259) else None
    bos = bos_token_id if bos_token_id is not None and (bos_token_id < 3 or bos_token_id >= 259) else None
    return ByteTokenizer(max(vocab_size, 259), eos_token_id=eos, bos_token_id=bos)


def build_prompt(tokenizer, user_message: str, system_prompt: str, mode: str = "auto") -> str:
    """mode: "zephyr" = the reference's fixed template; "hf" = the tokenizer's own chat template;
    "none" = raw text; "auto" = hf when the tokenizer ships one, else zephyr."""
    if mode == "none":
        return user_message
    if mode == "hf" or (mode == "auto" and getattr(tokenizer, "chat_template", None) and hasattr(tokenizer, "apply_chat_template")):
        try:
            return tokenizer.apply_chat_template(user_message, system_prompt)
        except Exception:
            if mode == "hf":
                raise
    return format_chat_prompt(user_message, system_prompt)


def build_chat_prompt(tokenizer, messages, default_system: str, mode: str = "auto") -> str:
    """A list of {"role": "system" | "user" | "assistant", "content": str} as one prompt that ends
    where the assistant's next turn begins: the tokenizer's own chat template when it ships one
    (mode "hf" / "auto"), else the Zephyr layout of the reference (orchestration.py:66) extended to
    several turns. Without a system message `default_system` is used, as the reference does."""
    msgs = [{"role": str(m.get("role", "user")), "content": str(m.get("content", ""))} for m in messages]
    if not any(m["role"] == "system" for m in msgs):
        msgs.insert(0, {"role": "system", "content": default_system})
    if mode == "none":
        return "\n".join(m["content"] for m in msgs if m["role"] != "system")
    tk = getattr(tokenizer, "tk", None)
    if mode in ("hf", "auto") and getattr(tokenizer, "chat_template", None) and tk is not None:
        try:
            return tk.apply_chat_template(msgs, tokenize=False, add_generation_prompt=True)
        except Exception:
            if mode == "hf":
                raise
    return "".join(f"<|{m['role']}|>\n{m['content']}</s>\n" for m in msgs) + "<|assistant|>\n"


def encode_prompt(tokenizer, text: str) -> List[int]:
    """Token ids of a formatted prompt with exactly one BOS: a chat template that already begins
    with the BOS string (Llama-3-Instruct's `<|begin_of_text|>`) is encoded without the tokenizer
    adding another one. The reference's Zephyr template carries none and gets the usual one
    (orchestration.py:98 encodes with the tokenizer's defaults)."""
    bos = getattr(tokenizer, "bos_token", None)
    if bos and text.startswith(bos):
        return tokenizer.encode(text, add_special_tokens=False)
    return tokenizer.encode(text)


class IncrementalDetokenizer:
    """Text of a growing token list, piece by piece, for streaming. Tokens [lo, mid) have been handed
    out; the text of [lo, n) minus the text of [lo, mid) is what is new. Both are decoded from the same
    start, so whatever the tokenizer does at the start of a string (dropping a leading space) cancels
    out, and a step costs two decodes of a few tokens, not one of the whole output. A token that
    ends inside a multi-byte character decodes to U+FFFD: the piece is held back until the rest has
    arrived (at most 4 tokens: by then it is not an unfinished character but invalid output)."""

    def __init__(self, tokenizer):
        self.tok = tokenizer
        self.lo = self.mid = 0

    def push(self, ids: Sequence[int], final: bool = False) -> str:
        """`ids`: all output tokens so far. Returns the new text ("" when nothing can be shown yet)."""
        n = len(ids)
        if n <= self.mid:
            return ""
        old = self.tok.decode(ids[self.lo:self.mid], skip_special_tokens=True)
        text = self.tok.decode(ids[self.lo:n], skip_special_tokens=True)
        if len(text) > len(old) and (final or not text.endswith("\ufffd") or n - self.mid >= 4):
            self.lo, self.mid = self.mid, n
            return text[len(old):]
        return ""

