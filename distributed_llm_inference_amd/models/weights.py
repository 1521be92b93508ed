"""Per-stage weights in the engine's logical layout.

Logical layout (all matrices `[out_features, in_features]`, i.e. what ``nn.Linear`` stores):

  embed        [V, H]            stage 0 (and the last stage when lm_head is tied)
  pos_embed    [P, H]            GPT-2 only, stage 0
  layers[i]:   ln1_w (+ln1_b)    [H]
               w_qkv (+b_qkv)    [Hq*D + 2*Hkv*D, H]     q,k,v concatenated -> one GEMM
               w_o   (+b_o)      [H, Hq*D]
               ln2_w (+ln2_b)    [H]
               w_gate, w_up      [I, H]  (Llama)   |  w_fc (+b_fc) [I, H]  (GPT-2)
               w_down (+b_down)  [H, I]
  final_norm_w (+final_norm_b)   [H]     last stage
  lm_head      [V, H]            last stage (aliases `embed` data when tied)

A stage materialises only what its `StagePlan` owns.  The reference instead loads the full
model in every process (orchestration.py:39-47, Worker1.py:60-75).

HF tensor names honoured — Llama: ``model.embed_tokens.weight``,
``model.layers.{i}.{input_layernorm,post_attention_layernorm}.weight``,
``model.layers.{i}.self_attn.{q,k,v,o}_proj.weight``,
``model.layers.{i}.mlp.{gate,up,down}_proj.weight``, ``model.norm.weight``, ``lm_head.weight``.
GPT-2 (optionally prefixed ``transformer.``): ``wte``, ``wpe``, ``h.{i}.ln_{1,2}``,
``h.{i}.attn.c_attn`` / ``c_proj``, ``h.{i}.mlp.c_fc`` / ``c_proj`` (Conv1D: stored
``[in, out]`` and therefore transposed here), ``ln_f``.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from .partition import StagePlan
from .safetensors_io import CheckpointReader
from .spec import ModelSpec


@dataclass
class StageWeights:
    spec: ModelSpec
    plan: StagePlan
    embed: Optional[torch.Tensor] = None
    pos_embed: Optional[torch.Tensor] = None
    layers: List[Dict[str, torch.Tensor]] = field(default_factory=list)
    final_norm_w: Optional[torch.Tensor] = None
    final_norm_b: Optional[torch.Tensor] = None
    lm_head: Optional[torch.Tensor] = None

    def num_bytes(self) -> int:
        seen, total = set(), 0
        for t in self.tensors():
            p = t.untyped_storage().data_ptr()
            if p in seen:
                continue
            seen.add(p)
            total += t.numel() * t.element_size()
        return total

    def tensors(self):
        for t in (self.embed, self.pos_embed, self.final_norm_w, self.final_norm_b, self.lm_head):
            if t is not None:
                yield t
        for l in self.layers:
            yield from l.values()

    def to(self, device=None, dtype=None) -> "StageWeights":
        def cv(t):
            if t is None:
                return None
            return t.to(device=device, dtype=dtype if t.is_floating_point() else None)
        tied = self.lm_head is not None and self.embed is not None and \
            self.lm_head.data_ptr() == self.embed.data_ptr()
        out = StageWeights(self.spec, self.plan)
        out.embed = cv(self.embed)
        out.pos_embed = cv(self.pos_embed)
        out.layers = [{k: cv(v) for k, v in l.items()} for l in self.layers]
        out.final_norm_w = cv(self.final_norm_w)
        out.final_norm_b = cv(self.final_norm_b)
        out.lm_head = out.embed if tied else cv(self.lm_head)
        return out


# --------------------------------------------------------------------------------------
# random initialisation (benchmarks and tests: no checkpoint, no network)
# --------------------------------------------------------------------------------------

def _seed_for(name: str, seed: int) -> int:
    h = hashlib.blake2b(f"{seed}:{name}".encode(), digest_size=8).digest()
    return int.from_bytes(h, "little") & 0x7FFF_FFFF_FFFF_FFFF


def _randn(name, shape, seed, device, dtype, std=0.02):
    """Deterministic per-tensor init: the value of a tensor depends only on (seed, name),
    never on which stage materialises it, so any partition of the model sees the same weights."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(_seed_for(name, seed))
    gen_dtype = torch.float32 if dev.type == "cpu" else dtype
    t = torch.empty(shape, device=dev, dtype=gen_dtype)
    t.normal_(0.0, std, generator=g)
    return t.to(dtype)


def random_stage_weights(spec: ModelSpec, plan: StagePlan, seed: int = 0, device="cpu",
                         dtype=torch.float32) -> StageWeights:
    """Random-init weights of the given architecture for one stage."""
    H, I, V = spec.hidden_size, spec.intermediate_size, spec.vocab_size
    w = StageWeights(spec, plan)
    need_embed = plan.has_embed or (plan.has_head and spec.tie_embeddings)
    if need_embed:
        w.embed = _randn("embed", (V, H), seed, device, dtype)
    if plan.has_embed and spec.is_gpt2:
        w.pos_embed = _randn("pos_embed", (spec.max_position, H), seed, device, dtype, std=0.01)
    ones = lambda n: torch.ones(n, device=device, dtype=dtype)
    for i in plan.layers:
        p = f"layers.{i}."
        l: Dict[str, torch.Tensor] = {}
        attn, mlp = plan.has_attn(i), plan.has_mlp(i)       # a stage boundary may fall between a layer's two blocks
        if attn:
            l["ln1_w"] = ones(H) + _randn(p + "ln1_w", (H,), seed, device, dtype, std=0.05)
            l["w_qkv"] = _randn(p + "w_qkv", (spec.qkv_size, H), seed, device, dtype)
            l["w_o"] = _randn(p + "w_o", (H, spec.q_size), seed, device, dtype)
        if mlp:
            l["ln2_w"] = ones(H) + _randn(p + "ln2_w", (H,), seed, device, dtype, std=0.05)
        if spec.is_llama:
            # (as large as the projections' outputs, like the q/k/v biases of trained Qwen2 checkpoints: a numerics
            # test then fails when a path drops or misplaces a bias)
            if attn and spec.qkv_bias:
                l["b_qkv"] = _randn(p + "b_qkv", (spec.qkv_size,), seed, device, dtype, std=0.3)
            if attn and spec.o_bias:
                l["b_o"] = _randn(p + "b_o", (H,), seed, device, dtype, std=0.3)
            if mlp:
                l["w_gate"] = _randn(p + "w_gate", (I, H), seed, device, dtype)
                l["w_up"] = _randn(p + "w_up", (I, H), seed, device, dtype)
                l["w_down"] = _randn(p + "w_down", (H, I), seed, device, dtype)
        else:
            if attn:
                l["ln1_b"] = _randn(p + "ln1_b", (H,), seed, device, dtype)
                l["b_qkv"] = _randn(p + "b_qkv", (spec.qkv_size,), seed, device, dtype)
                l["b_o"] = _randn(p + "b_o", (H,), seed, device, dtype)
            if mlp:
                l["ln2_b"] = _randn(p + "ln2_b", (H,), seed, device, dtype)
                l["w_fc"] = _randn(p + "w_fc", (I, H), seed, device, dtype)
                l["b_fc"] = _randn(p + "b_fc", (I,), seed, device, dtype)
                l["w_down"] = _randn(p + "w_down", (H, I), seed, device, dtype)
                l["b_down"] = _randn(p + "b_down", (H,), seed, device, dtype)
        w.layers.append(l)
    if plan.has_head:
        w.final_norm_w = ones(H) + _randn("final_norm_w", (H,), seed, device, dtype, std=0.05)
        if spec.is_gpt2:
            w.final_norm_b = _randn("final_norm_b", (H,), seed, device, dtype)
        w.lm_head = w.embed if spec.tie_embeddings else _randn("lm_head", (V, H), seed, device, dtype)
    return w


# --------------------------------------------------------------------------------------
# Hugging Face checkpoint -> stage weights
# --------------------------------------------------------------------------------------

def stage_tensor_names(spec: ModelSpec, plan: StagePlan, available=None) -> List[str]:
    """The HF tensor names a stage needs (used by tests to prove nothing else is touched)."""
    names: List[str] = []
    if spec.is_llama:
        if plan.has_embed or (plan.has_head and spec.tie_embeddings):
            names.append("model.embed_tokens.weight")
        for i in plan.layers:
            p = f"model.layers.{i}."
            if plan.has_attn(i):
                names += [p + "input_layernorm.weight"] + [p + f"self_attn.{x}_proj.weight" for x in "qkvo"]
                names += [p + f"self_attn.{x}_proj.bias" for x in ("qkv" if spec.qkv_bias else "") + ("o" if spec.o_bias else "")]
            if plan.has_mlp(i):
                names += [p + "post_attention_layernorm.weight"] + [p + f"mlp.{x}_proj.weight" for x in ("gate", "up", "down")]
        if plan.has_head:
            names.append("model.norm.weight")
            if not spec.tie_embeddings:
                names.append("lm_head.weight")
    else:
        if plan.has_embed or plan.has_head:
            names.append("wte.weight")
        if plan.has_embed:
            names.append("wpe.weight")
        for i in plan.layers:
            p = f"h.{i}."
            for m in (("ln_1", "attn.c_attn", "attn.c_proj") if plan.has_attn(i) else ()) + \
                     (("ln_2", "mlp.c_fc", "mlp.c_proj") if plan.has_mlp(i) else ()):
                names += [p + m + ".weight", p + m + ".bias"]
        if plan.has_head:
            names += ["ln_f.weight", "ln_f.bias"]
    return names


def load_stage_weights(model_dir: str, spec: ModelSpec, plan: StagePlan, device="cpu",
                       dtype=torch.float32, reader: Optional[CheckpointReader] = None) -> StageWeights:
    """Read exactly this stage's tensors from a HF checkpoint directory."""
    rd = reader or CheckpointReader(model_dir)
    w = StageWeights(spec, plan)

    def get(name):
        t = rd.get(name)
        return t.to(device=device, dtype=dtype)

    if spec.is_llama:
        has_lm_head = "lm_head.weight" in rd
        tied = spec.tie_embeddings or not has_lm_head
        if plan.has_embed or (plan.has_head and tied):
            w.embed = get("model.embed_tokens.weight")
        for i in plan.layers:
            p = f"model.layers.{i}."
            l = {}
            if plan.has_attn(i):
                l.update({
                    "ln1_w": get(p + "input_layernorm.weight"),
                    "w_qkv": torch.cat([get(p + "self_attn.q_proj.weight"),
                                        get(p + "self_attn.k_proj.weight"),
                                        get(p + "self_attn.v_proj.weight")], dim=0),
                    "w_o": get(p + "self_attn.o_proj.weight")})
                if spec.qkv_bias:
                    l["b_qkv"] = torch.cat([get(p + f"self_attn.{x}_proj.bias") for x in "qkv"], dim=0)
                if spec.o_bias:
                    l["b_o"] = get(p + "self_attn.o_proj.bias")
            if plan.has_mlp(i):
                l.update({
                    "ln2_w": get(p + "post_attention_layernorm.weight"),
                    "w_gate": get(p + "mlp.gate_proj.weight"),
                    "w_up": get(p + "mlp.up_proj.weight"),
                    "w_down": get(p + "mlp.down_proj.weight")})
            w.layers.append(l)
        if plan.has_head:
            w.final_norm_w = get("model.norm.weight")
            w.lm_head = w.embed if tied else get("lm_head.weight")
    else:
        pre = "transformer." if "transformer.wte.weight" in rd else ""
        if plan.has_embed or plan.has_head:
            w.embed = get(pre + "wte.weight")
        if plan.has_embed:
            w.pos_embed = get(pre + "wpe.weight")
        for i in plan.layers:
            p = f"{pre}h.{i}."
            # Conv1D stores [in, out]; the engine wants [out, in]
            l = {}
            if plan.has_attn(i):
                l.update({
                    "ln1_w": get(p + "ln_1.weight"), "ln1_b": get(p + "ln_1.bias"),
                    "w_qkv": get(p + "attn.c_attn.weight").t().contiguous(), "b_qkv": get(p + "attn.c_attn.bias"),
                    "w_o": get(p + "attn.c_proj.weight").t().contiguous(), "b_o": get(p + "attn.c_proj.bias")})
            if plan.has_mlp(i):
                l.update({
                    "ln2_w": get(p + "ln_2.weight"), "ln2_b": get(p + "ln_2.bias"),
                    "w_fc": get(p + "mlp.c_fc.weight").t().contiguous(), "b_fc": get(p + "mlp.c_fc.bias"),
                    "w_down": get(p + "mlp.c_proj.weight").t().contiguous(), "b_down": get(p + "mlp.c_proj.bias")})
            w.layers.append(l)
        if plan.has_head:
            w.final_norm_w = get(pre + "ln_f.weight")
            w.final_norm_b = get(pre + "ln_f.bias")
            w.lm_head = w.embed
    if reader is None:
        rd.close()
    return w


def export_hf_state_dict(parts: List[StageWeights]) -> Dict[str, torch.Tensor]:
    """Inverse mapping (logical -> HF names) over all stages; used by tests to build HF models
    holding exactly the engine's random weights."""
    spec = parts[0].spec
    sd: Dict[str, torch.Tensor] = {}
    for w in parts:
        if spec.is_llama:
            if w.plan.has_embed:
                sd["model.embed_tokens.weight"] = w.embed
            for i, l in zip(w.plan.layers, w.layers):
                p = f"model.layers.{i}."
                if "w_qkv" in l:
                    q, k, v = torch.split(l["w_qkv"], [spec.q_size, spec.kv_size, spec.kv_size], dim=0)
                    sd[p + "input_layernorm.weight"] = l["ln1_w"]
                    sd[p + "self_attn.q_proj.weight"] = q
                    sd[p + "self_attn.k_proj.weight"] = k
                    sd[p + "self_attn.v_proj.weight"] = v
                    sd[p + "self_attn.o_proj.weight"] = l["w_o"]
                    if "b_qkv" in l:
                        for x, b in zip("qkv", torch.split(l["b_qkv"], [spec.q_size, spec.kv_size, spec.kv_size], dim=0)):
                            sd[p + f"self_attn.{x}_proj.bias"] = b
                    if "b_o" in l:
                        sd[p + "self_attn.o_proj.bias"] = l["b_o"]
                if "w_down" in l:
                    sd[p + "post_attention_layernorm.weight"] = l["ln2_w"]
                    sd[p + "mlp.gate_proj.weight"] = l["w_gate"]
                    sd[p + "mlp.up_proj.weight"] = l["w_up"]
                    sd[p + "mlp.down_proj.weight"] = l["w_down"]
            if w.plan.has_head:
                sd["model.norm.weight"] = w.final_norm_w
                if not spec.tie_embeddings:
                    sd["lm_head.weight"] = w.lm_head
        else:
            pre = "transformer."
            if w.plan.has_embed:
                sd[pre + "wte.weight"] = w.embed
                sd[pre + "wpe.weight"] = w.pos_embed
            for i, l in zip(w.plan.layers, w.layers):
                p = f"{pre}h.{i}."
                if "w_qkv" in l:
                    sd[p + "ln_1.weight"], sd[p + "ln_1.bias"] = l["ln1_w"], l["ln1_b"]
                    sd[p + "attn.c_attn.weight"], sd[p + "attn.c_attn.bias"] = l["w_qkv"].t(), l["b_qkv"]
                    sd[p + "attn.c_proj.weight"], sd[p + "attn.c_proj.bias"] = l["w_o"].t(), l["b_o"]
                if "w_down" in l:
                    sd[p + "ln_2.weight"], sd[p + "ln_2.bias"] = l["ln2_w"], l["ln2_b"]
                    sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"] = l["w_fc"].t(), l["b_fc"]
                    sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"] = l["w_down"].t(), l["b_down"]
            if w.plan.has_head:
                sd[pre + "ln_f.weight"], sd[pre + "ln_f.bias"] = w.final_norm_w, w.final_norm_b
    return {k: v.detach().clone().contiguous() for k, v in sd.items()}
