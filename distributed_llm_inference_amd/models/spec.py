"""Architecture description of a decoder-only transformer.

A `ModelSpec` is everything the engine needs to know about a checkpoint's
architecture: it is parsed from a Hugging Face ``config.json`` (Llama family —
Llama, Mistral, Qwen2 — and GPT-2 family) or taken from the built-in table of
benchmark models.

The reference hard-codes one model name as a module constant and leans on
``AutoModelForCausalLM`` for everything else (orchestration.py:20,39-47;
Worker1.py:26,60-70).  Here the architecture is explicit so that each pipeline
stage can allocate exactly its own layers, KV cache and RoPE tables.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, asdict, replace
from typing import Optional, Dict, Any


@dataclass(frozen=True)
class RopeScaling:
    """``rope_scaling`` of a config.json. ``llama3``: Llama-3.1 frequency-dependent rescale (uses
    ``low_freq_factor`` / ``high_freq_factor``). ``linear``: positions divided by ``factor``. ``yarn``:
    interpolated below / extrapolated above a band of frequencies set by ``beta_fast`` / ``beta_slow``,
    cos and sin scaled by ``attention_factor`` (default 0.1 ln(factor) + 1). All of them are static, i.e. a
    table per stage; ``dynamic`` (a base that depends on the running sequence length) is refused."""
    factor: float = 8.0
    low_freq_factor: float = 1.0
    high_freq_factor: float = 4.0
    original_max_position_embeddings: int = 8192
    rope_type: str = "llama3"
    beta_fast: float = 32.0
    beta_slow: float = 1.0
    attention_factor: Optional[float] = None
    truncate: bool = True


@dataclass(frozen=True)
class ModelSpec:
    name: str
    family: str                     # "llama" | "gpt2"
    num_layers: int
    hidden_size: int
    num_heads: int
    num_kv_heads: int
    head_dim: int
    intermediate_size: int
    vocab_size: int
    max_position: int
    norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    rope_scaling: Optional[RopeScaling] = None
    tie_embeddings: bool = False
    bos_token_id: Optional[int] = None
    eos_token_id: Optional[int] = None
    # further end-of-sequence ids (config.json / generation_config.json may list several: Llama-3-Instruct ends a
    # turn with <|eot_id|>, not with the id listed first)
    extra_eos_token_ids: tuple = ()
    # Llama-family variants: which config.json dialect the spec came from ("llama" | "mistral" | "qwen2"),
    # biases on the q/k/v projections (Qwen2; Llama with `attention_bias`) and on the output projection
    # (Llama with `attention_bias`), and the attention window of checkpoints trained with one (Mistral)
    hf_type: str = "llama"
    qkv_bias: bool = False
    o_bias: bool = False
    sliding_window: Optional[int] = None

    def check_context(self, max_seq_len: int):
        """Attention here is over the whole context. A checkpoint trained with a sliding window gives
        the same results only while no sequence is longer than the window."""
        if self.sliding_window is not None and max_seq_len > self.sliding_window:
            raise ValueError(f"{self.name}: sliding-window attention (window {self.sliding_window}) is not implemented; "
                             f"contexts up to the window are exact, max_seq_len={max_seq_len} is beyond it")

    # ---- derived quantities -------------------------------------------------
    @property
    def q_size(self) -> int:
        return self.num_heads * self.head_dim

    @property
    def kv_size(self) -> int:
        return self.num_kv_heads * self.head_dim

    @property
    def qkv_size(self) -> int:
        return self.q_size + 2 * self.kv_size

    @property
    def group_size(self) -> int:
        return self.num_heads // self.num_kv_heads

    @property
    def is_llama(self) -> bool:
        return self.family == "llama"

    @property
    def is_gpt2(self) -> bool:
        return self.family == "gpt2"

    def layer_param_count(self) -> int:
        h, i = self.hidden_size, self.intermediate_size
        if self.is_llama:
            return h * self.qkv_size + self.q_size * h + 3 * h * i + 2 * h + self._attn_bias_count()
        # GPT-2: c_attn, c_proj, c_fc, mlp c_proj, all with biases, two LayerNorms
        return (h * 3 * h + 3 * h) + (h * h + h) + (h * i + i) + (i * h + h) + 4 * h

    def attn_param_count(self) -> int:
        """Parameters of a layer's attention block (its norm, QKV and output projections)."""
        h = self.hidden_size
        if self.is_llama:
            return h * self.qkv_size + self.q_size * h + h + self._attn_bias_count()
        return (h * 3 * h + 3 * h) + (h * h + h) + 2 * h

    def _attn_bias_count(self) -> int:
        return (self.qkv_size if self.qkv_bias else 0) + (self.hidden_size if self.o_bias else 0)

    def embed_param_count(self) -> int:
        n = self.vocab_size * self.hidden_size
        if self.is_gpt2:
            n += self.max_position * self.hidden_size
        return n

    def head_param_count(self) -> int:
        """Final norm + lm_head (the lm_head matrix is read every step even when tied)."""
        return self.vocab_size * self.hidden_size + self.hidden_size * (2 if self.is_gpt2 else 1)

    def param_count(self) -> int:
        n = self.num_layers * self.layer_param_count() + self.embed_param_count()
        n += self.hidden_size * (2 if self.is_gpt2 else 1)
        if not self.tie_embeddings:
            n += self.vocab_size * self.hidden_size
        return n

    def kv_bytes_per_token_per_layer(self, dtype_bytes: int = 2) -> int:
        return 2 * self.num_kv_heads * self.head_dim * dtype_bytes

    def to_dict(self) -> Dict[str, Any]:
        d = asdict(self)
        return d

    # ---- constructors ------------------------------------------------------
    @staticmethod
    def from_hf_config(cfg: Dict[str, Any], name: str = "") -> "ModelSpec":
        """Build a spec from the dict form of a Hugging Face ``config.json``."""
        mt = cfg.get("model_type", "")
        archs = cfg.get("architectures") or []
        if mt == "gpt2" or any("GPT2" in a for a in archs):
            if cfg.get("activation_function", "gelu_new") != "gelu_new" or cfg.get("scale_attn_by_inverse_layer_idx"):
                raise ValueError("GPT-2 variants other than gelu_new MLPs with plain 1/sqrt(d) attention scaling are not supported")
            h = cfg.get("n_embd", cfg.get("hidden_size"))
            nh = cfg.get("n_head", cfg.get("num_attention_heads"))
            inner = cfg.get("n_inner") or 4 * h
            return ModelSpec(
                name=name or cfg.get("_name_or_path", "gpt2"),
                family="gpt2",
                num_layers=cfg.get("n_layer", cfg.get("num_hidden_layers")),
                hidden_size=h,
                num_heads=nh,
                num_kv_heads=nh,
                head_dim=h // nh,
                intermediate_size=inner,
                vocab_size=cfg["vocab_size"],
                max_position=cfg.get("n_positions", cfg.get("max_position_embeddings", 1024)),
                norm_eps=cfg.get("layer_norm_epsilon", 1e-5),
                tie_embeddings=True,
                bos_token_id=cfg.get("bos_token_id"),
                eos_token_id=_scalar_eos(cfg.get("eos_token_id")),
            )
        # (a config.json without model_type that has the Llama keys is taken as Llama; a NAMED other architecture is
        # refused rather than run as if it were one: Gemma, Phi, ... differ in norms, activations and scalings)
        if mt in ("llama", "mistral", "qwen2") or any(a.startswith(("Llama", "Mistral", "Qwen2For")) for a in archs) \
                or (not mt and not archs and "num_key_value_heads" in cfg):
            if cfg.get("hidden_act", "silu") not in ("silu", "swish"):
                raise ValueError(f"hidden_act {cfg.get('hidden_act')!r} is not supported (the fused MLP epilogue is SiLU-gated)")
            if float(cfg.get("partial_rotary_factor", 1.0)) != 1.0:
                raise ValueError("partial rotary embeddings are not supported")
            h = cfg["hidden_size"]
            nh = cfg["num_attention_heads"]
            rs = cfg.get("rope_scaling")
            rope_theta = cfg.get("rope_theta")
            # transformers >= 5 nests theta/scaling under `rope_parameters`
            rp = cfg.get("rope_parameters")
            if isinstance(rp, dict):
                rope_theta = rp.get("rope_theta", rope_theta)
                if rp.get("rope_type", "default") not in ("default", None):
                    rs = rp
            scaling = None
            rtype = rs.get("rope_type", rs.get("type")) if isinstance(rs, dict) else None
            if rtype == "llama3":
                scaling = RopeScaling(
                    factor=float(rs.get("factor", 8.0)),
                    low_freq_factor=float(rs.get("low_freq_factor", 1.0)),
                    high_freq_factor=float(rs.get("high_freq_factor", 4.0)),
                    original_max_position_embeddings=int(rs.get("original_max_position_embeddings", 8192)),
                )
            elif rtype == "linear":
                scaling = RopeScaling(factor=float(rs["factor"]), rope_type="linear",
                                      original_max_position_embeddings=int(cfg.get("max_position_embeddings", 2048)))
            elif rtype == "yarn":
                if rs.get("mscale") or rs.get("mscale_all_dim"):
                    raise ValueError("yarn rope_scaling with mscale / mscale_all_dim is not supported")
                orig = int(rs.get("original_max_position_embeddings") or cfg.get("max_position_embeddings", 2048))
                factor = rs.get("factor")
                scaling = RopeScaling(
                    factor=float(factor if factor is not None else cfg.get("max_position_embeddings", orig) / orig),
                    rope_type="yarn", original_max_position_embeddings=orig,
                    beta_fast=float(rs.get("beta_fast") or 32.0), beta_slow=float(rs.get("beta_slow") or 1.0),
                    attention_factor=(float(rs["attention_factor"]) if rs.get("attention_factor") is not None else None),
                    truncate=bool(rs.get("truncate", True)))
            elif rtype not in (None, "default"):
                raise ValueError(f"rope_scaling type {rtype!r} is not supported (llama3, linear and yarn are)")
            if cfg.get("mlp_bias"):
                raise ValueError("mlp_bias=true is not supported (the fused gate/up epilogue has no bias)")
            hf_type = mt if mt in ("mistral", "qwen2") else "llama"
            attn_bias = bool(cfg.get("attention_bias", False))
            window = cfg.get("sliding_window")
            if hf_type == "qwen2" and not cfg.get("use_sliding_window", False):
                window = None
            if hf_type == "llama":
                window = None
            return ModelSpec(
                name=name or cfg.get("_name_or_path", "llama"),
                family="llama",
                num_layers=cfg["num_hidden_layers"],
                hidden_size=h,
                num_heads=nh,
                num_kv_heads=cfg.get("num_key_value_heads", nh),
                head_dim=cfg.get("head_dim") or h // nh,
                intermediate_size=cfg["intermediate_size"],
                vocab_size=cfg["vocab_size"],
                max_position=cfg.get("max_position_embeddings", 2048),
                norm_eps=cfg.get("rms_norm_eps", 1e-5),
                rope_theta=float(rope_theta if rope_theta is not None else 10000.0),
                rope_scaling=scaling,
                tie_embeddings=bool(cfg.get("tie_word_embeddings", False)),
                bos_token_id=cfg.get("bos_token_id"),
                eos_token_id=_scalar_eos(cfg.get("eos_token_id")),
                extra_eos_token_ids=_more_eos(cfg.get("eos_token_id")),
                hf_type=hf_type,
                qkv_bias=True if hf_type == "qwen2" else attn_bias,       # Qwen2: q/k/v always biased, o never
                o_bias=False if hf_type == "qwen2" else attn_bias,
                sliding_window=int(window) if window else None,
            )
        raise ValueError(f"unsupported model_type {mt!r} (architectures={archs})")

    @staticmethod
    def from_pretrained(path_or_name: str) -> "ModelSpec":
        """Resolve a built-in model name or a local directory holding ``config.json``."""
        key = path_or_name.lower()
        if key in BUILTIN_SPECS:
            return BUILTIN_SPECS[key]
        short = key.split("/")[-1]
        if short in BUILTIN_SPECS and not os.path.isdir(path_or_name):
            return BUILTIN_SPECS[short]
        cfg_path = os.path.join(path_or_name, "config.json")
        if os.path.isfile(cfg_path):
            with open(cfg_path) as f:
                spec = ModelSpec.from_hf_config(json.load(f), name=path_or_name)
            # generation_config.json may name more end-of-sequence ids than config.json does
            gen_path = os.path.join(path_or_name, "generation_config.json")
            if os.path.isfile(gen_path):
                with open(gen_path) as f:
                    ge = json.load(f).get("eos_token_id")
                ids = [int(x) for x in (ge if isinstance(ge, (list, tuple)) else [ge] if ge is not None else [])]
                more = tuple(dict.fromkeys(x for x in (*spec.extra_eos_token_ids, *ids) if x != spec.eos_token_id))
                if spec.eos_token_id is None and more:
                    spec = replace(spec, eos_token_id=more[0], extra_eos_token_ids=more[1:])
                else:
                    spec = replace(spec, extra_eos_token_ids=more)
            return spec
        raise FileNotFoundError(
            f"{path_or_name!r} is neither a built-in model ({sorted(BUILTIN_SPECS)}) "
            f"nor a directory with config.json (the Hugging Face hub is not reachable)")

    def hf_config(self) -> Dict[str, Any]:
        """The inverse of `from_hf_config`: a config.json dict for this spec."""
        if self.is_gpt2:
            return {
                "model_type": "gpt2", "architectures": ["GPT2LMHeadModel"],
                "n_layer": self.num_layers, "n_embd": self.hidden_size, "n_head": self.num_heads,
                "n_inner": self.intermediate_size, "vocab_size": self.vocab_size,
                "n_positions": self.max_position, "layer_norm_epsilon": self.norm_eps,
                "activation_function": "gelu_new",
                "bos_token_id": self.bos_token_id, "eos_token_id": self.eos_token_id,
            }
        cfg = {
            "model_type": "llama", "architectures": ["LlamaForCausalLM"],
            "num_hidden_layers": self.num_layers, "hidden_size": self.hidden_size,
            "num_attention_heads": self.num_heads, "num_key_value_heads": self.num_kv_heads,
            "head_dim": self.head_dim, "intermediate_size": self.intermediate_size,
            "vocab_size": self.vocab_size, "max_position_embeddings": self.max_position,
            "rms_norm_eps": self.norm_eps, "rope_theta": self.rope_theta,
            "tie_word_embeddings": self.tie_embeddings, "hidden_act": "silu",
            "attention_bias": self.qkv_bias, "mlp_bias": False,
            "bos_token_id": self.bos_token_id, "eos_token_id": self.eos_token_id,
        }
        if self.hf_type == "qwen2":
            cfg.update({"model_type": "qwen2", "architectures": ["Qwen2ForCausalLM"],
                        "use_sliding_window": self.sliding_window is not None, "sliding_window": self.sliding_window,
                        "max_window_layers": 0})
            del cfg["attention_bias"], cfg["mlp_bias"], cfg["head_dim"]
        elif self.hf_type == "mistral":
            cfg.update({"model_type": "mistral", "architectures": ["MistralForCausalLM"],
                        "sliding_window": self.sliding_window})
            del cfg["attention_bias"], cfg["mlp_bias"]
        elif self.qkv_bias != self.o_bias:
            raise ValueError("a Llama config.json has one `attention_bias` for q/k/v and o alike")
        if self.extra_eos_token_ids:
            cfg["eos_token_id"] = [self.eos_token_id, *self.extra_eos_token_ids]
        if self.rope_scaling is not None:
            rs = self.rope_scaling
            if rs.rope_type == "linear":
                cfg["rope_scaling"] = {"rope_type": "linear", "factor": rs.factor}
            elif rs.rope_type == "yarn":
                cfg["rope_scaling"] = {"rope_type": "yarn", "factor": rs.factor, "beta_fast": rs.beta_fast, "beta_slow": rs.beta_slow,
                                       "original_max_position_embeddings": rs.original_max_position_embeddings,
                                       "truncate": rs.truncate}
                if rs.attention_factor is not None:
                    cfg["rope_scaling"]["attention_factor"] = rs.attention_factor
            else:
                cfg["rope_scaling"] = {
                    "rope_type": "llama3", "factor": rs.factor, "low_freq_factor": rs.low_freq_factor,
                    "high_freq_factor": rs.high_freq_factor,
                    "original_max_position_embeddings": rs.original_max_position_embeddings,
                }
        return cfg


def _scalar_eos(v):
    if isinstance(v, (list, tuple)):
        return v[0] if v else None
    return v


def _more_eos(v) -> tuple:
    return tuple(int(x) for x in v[1:]) if isinstance(v, (list, tuple)) else ()


# Shapes of the benchmark models.  The reference serves TinyLlama-1.1B-Chat
# (orchestration.py:20); GPT-2 small and Llama-3 8B/70B are the other benchmark shapes.
BUILTIN_SPECS: Dict[str, ModelSpec] = {
    "gpt2": ModelSpec(
        name="gpt2", family="gpt2", num_layers=12, hidden_size=768, num_heads=12, num_kv_heads=12,
        head_dim=64, intermediate_size=3072, vocab_size=50257, max_position=1024, norm_eps=1e-5,
        tie_embeddings=True, bos_token_id=50256, eos_token_id=50256),
    "tinyllama-1.1b": ModelSpec(
        name="TinyLlama/TinyLlama-1.1B-Chat-v1.0", family="llama", num_layers=22, hidden_size=2048,
        num_heads=32, num_kv_heads=4, head_dim=64, intermediate_size=5632, vocab_size=32000,
        max_position=2048, norm_eps=1e-5, rope_theta=10000.0, bos_token_id=1, eos_token_id=2),
    "llama-3-8b": ModelSpec(
        name="meta-llama/Meta-Llama-3-8B", family="llama", num_layers=32, hidden_size=4096,
        num_heads=32, num_kv_heads=8, head_dim=128, intermediate_size=14336, vocab_size=128256,
        max_position=8192, norm_eps=1e-5, rope_theta=500000.0, bos_token_id=128000, eos_token_id=128001),
    "llama-3-70b": ModelSpec(
        name="meta-llama/Meta-Llama-3-70B", family="llama", num_layers=80, hidden_size=8192,
        num_heads=64, num_kv_heads=8, head_dim=128, intermediate_size=28672, vocab_size=128256,
        max_position=8192, norm_eps=1e-5, rope_theta=500000.0, bos_token_id=128000, eos_token_id=128001),
    # Llama-family variants: Mistral (trained with a 4096-token attention window: exact up to there) and
    # Qwen2 (biases on the q/k/v projections, tied embeddings at this size)
    "mistral-7b": ModelSpec(
        name="mistralai/Mistral-7B-v0.1", family="llama", num_layers=32, hidden_size=4096, num_heads=32,
        num_kv_heads=8, head_dim=128, intermediate_size=14336, vocab_size=32000, max_position=32768,
        norm_eps=1e-5, rope_theta=10000.0, bos_token_id=1, eos_token_id=2, hf_type="mistral", sliding_window=4096),
    "qwen2.5-3b": ModelSpec(
        name="Qwen/Qwen2.5-3B", family="llama", num_layers=36, hidden_size=2048, num_heads=16, num_kv_heads=2,
        head_dim=128, intermediate_size=11008, vocab_size=151936, max_position=32768, norm_eps=1e-6,
        rope_theta=1000000.0, tie_embeddings=True, bos_token_id=151643, eos_token_id=151643,
        hf_type="qwen2", qkv_bias=True),
    # GQA groups that are not powers of two: 3 query heads per kv head (Llama-3.2-3B), 7 (Qwen2.5-7B)
    "llama-3.2-3b": ModelSpec(
        name="meta-llama/Llama-3.2-3B", family="llama", num_layers=28, hidden_size=3072, num_heads=24, num_kv_heads=8,
        head_dim=128, intermediate_size=8192, vocab_size=128256, max_position=131072, norm_eps=1e-5,
        rope_theta=500000.0, rope_scaling=RopeScaling(32.0, 1.0, 4.0, 8192), tie_embeddings=True,
        bos_token_id=128000, eos_token_id=128001),
    "qwen2.5-7b": ModelSpec(
        name="Qwen/Qwen2.5-7B", family="llama", num_layers=28, hidden_size=3584, num_heads=28, num_kv_heads=4,
        head_dim=128, intermediate_size=18944, vocab_size=152064, max_position=131072, norm_eps=1e-6,
        rope_theta=1000000.0, bos_token_id=151643, eos_token_id=151643, hf_type="qwen2", qkv_bias=True),
}
# aliases
BUILTIN_SPECS["mistral-7b-v0.1"] = BUILTIN_SPECS["mistral-7b"]
BUILTIN_SPECS["tinyllama"] = BUILTIN_SPECS["tinyllama-1.1b"]
BUILTIN_SPECS["tinyllama-1.1b-chat-v1.0"] = BUILTIN_SPECS["tinyllama-1.1b"]
BUILTIN_SPECS["gpt2-small"] = BUILTIN_SPECS["gpt2"]
BUILTIN_SPECS["llama3-8b"] = BUILTIN_SPECS["llama-3-8b"]
BUILTIN_SPECS["meta-llama-3-8b"] = BUILTIN_SPECS["llama-3-8b"]
BUILTIN_SPECS["llama3-70b"] = BUILTIN_SPECS["llama-3-70b"]
BUILTIN_SPECS["meta-llama-3-70b"] = BUILTIN_SPECS["llama-3-70b"]


def _register_test_models():
    # resolvable by name in spawned stage processes too (tests of the multi-process pipeline)
    BUILTIN_SPECS["tiny-llama-test"] = tiny_spec("llama")
    BUILTIN_SPECS["tiny-gpt2-test"] = tiny_spec("gpt2")
    BUILTIN_SPECS["tiny-llama-8l-test"] = tiny_spec("llama", name="tiny-llama-8l-test", num_layers=8)   # splits into 8 stages
    BUILTIN_SPECS["tiny-llama-gpu-test"] = tiny_spec(
        "llama", name="tiny-llama-gpu-test", hidden_size=256, num_heads=4, num_kv_heads=2, head_dim=64,
        intermediate_size=512, vocab_size=320, num_layers=4)
    BUILTIN_SPECS["tiny-qwen2-gpu-test"] = tiny_spec(
        "llama", name="tiny-qwen2-gpu-test", hidden_size=256, num_heads=4, num_kv_heads=2, head_dim=64,
        intermediate_size=512, vocab_size=320, num_layers=4, hf_type="qwen2", qkv_bias=True, tie_embeddings=True)
    BUILTIN_SPECS["tiny-gpt2-gpu-test"] = tiny_spec(
        "gpt2", name="tiny-gpt2-gpu-test", hidden_size=256, num_heads=4, num_kv_heads=4, head_dim=64,
        intermediate_size=1024, vocab_size=333, num_layers=4)


def tiny_spec(family: str = "llama", **overrides) -> ModelSpec:
    """A toy architecture for unit tests (seconds on CPU)."""
    if family == "llama":
        base = dict(name="tiny-llama-test", family="llama", num_layers=4, hidden_size=128, num_heads=4,
                    num_kv_heads=2, head_dim=32, intermediate_size=256, vocab_size=320, max_position=256,
                    norm_eps=1e-5, rope_theta=10000.0, bos_token_id=1, eos_token_id=2)
    else:
        base = dict(name="tiny-gpt2-test", family="gpt2", num_layers=4, hidden_size=128, num_heads=4,
                    num_kv_heads=4, head_dim=32, intermediate_size=512, vocab_size=320, max_position=256,
                    norm_eps=1e-5, tie_embeddings=True, bos_token_id=0, eos_token_id=0)
    base.update(overrides)
    return ModelSpec(**base)


_register_test_models()
