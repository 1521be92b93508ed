"""Pure-PyTorch pipeline-stage executor (the golden model and the no-GPU backend).

Implements exactly the stage interface of the GPU executor — ragged `StepBatch` in, hidden
states or logits out, with a per-sequence KV cache — using plain torch ops, so the scheduler,
transports, API and every pipeline-equivalence property are testable on CPU, and so the HIP
path has an independent model to be compared with.

What a stage computes is what the reference's `Worker.process` computes for its layer range
(Worker1.py:82-177: position ids, RoPE, the decoder-layer loop), plus the orchestrator's head
and tail on the first/last stage (orchestration.py:111 embedding, :140-141 final norm +
lm_head) — except that only new tokens are processed (KV cache) and only the rows that will
be sampled go through the final norm and lm_head.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from ..ops import reference as R
from ..runtime.batch import StepBatch
from ..runtime.sampler import filter_logits
from .partition import StagePlan
from .rope import build_rope_table
from .spec import ModelSpec
from .weights import StageWeights


class RefStage:
    backend = "torch"

    def __init__(self, spec: ModelSpec, plan: StagePlan, weights: StageWeights, dtype=torch.float32,
                 device="cpu", max_seq_len: Optional[int] = None):
        self.spec, self.plan = spec, plan
        self.dtype, self.device = dtype, torch.device(device)
        self.w = weights.to(device=self.device, dtype=dtype)
        self.max_seq_len = max_seq_len or min(spec.max_position, spec.sliding_window or spec.max_position)
        spec.check_context(self.max_seq_len)
        if spec.is_llama:
            self.cos, self.sin = build_rope_table(spec, self.max_seq_len, self.device)
        # slot -> per-layer [K, V] tensors of shape [len, Hkv, D]
        self.cache: Dict[int, List[List[torch.Tensor]]] = {}

    # -- helpers --------------------------------------------------------------------------
    def _norm(self, x, w, b):
        if self.spec.is_llama:
            return R.rmsnorm(x, w, self.spec.norm_eps).to(self.dtype)
        return R.layernorm(x, w, b, self.spec.norm_eps).to(self.dtype)

    def _slot_cache(self, slot: int):
        c = self.cache.get(slot)
        if c is None:
            z = lambda: torch.zeros(0, self.spec.num_kv_heads, self.spec.head_dim, dtype=self.dtype, device=self.device)
            c = [[z(), z()] for _ in self.plan.layers]
            self.cache[slot] = c
        return c

    def release(self, slots: List[int]):
        for s in slots:
            self.cache.pop(s, None)

    # -- head / tail alone (the HTTP-compat orchestrator owns only these, as the reference's
    #    orchestrator does: orchestration.py:45-47) ---------------------------------------
    @torch.no_grad()
    def embed_tokens(self, ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        h = self.w.embed[ids.long()]
        if self.spec.is_gpt2:
            h = h + self.w.pos_embed[positions.long()]
        return h.to(self.dtype)

    @torch.no_grad()
    def head_logits(self, h_rows: torch.Tensor) -> torch.Tensor:
        hn = self._norm(h_rows.to(self.dtype), self.w.final_norm_w, self.w.final_norm_b)
        return R.gemm(hn, self.w.lm_head).float()

    # -- forward --------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, batch: StepBatch, x: torch.Tensor) -> torch.Tensor:
        spec = self.spec
        T = batch.num_tokens
        positions = torch.tensor([c + i for c, q in zip(batch.ctx_lens, batch.q_lens) for i in range(q)],
                                 dtype=torch.long, device=self.device)
        if self.plan.has_embed:
            ids = x.to(self.device).long().reshape(-1)
            assert ids.numel() == T, f"expected {T} token ids, got {ids.numel()}"
            h = self.w.embed[ids]
            if spec.is_gpt2:
                h = h + self.w.pos_embed[positions]
            h = h.to(self.dtype)
        else:
            h = x.to(device=self.device, dtype=self.dtype).reshape(T, spec.hidden_size)
        Hq, Hkv, D = spec.num_heads, spec.num_kv_heads, spec.head_dim
        scale = D ** -0.5
        for li, l in enumerate(self.w.layers):
          if "w_qkv" in l:                 # (a stage boundary may fall between a layer's attention and MLP blocks)
            xn = self._norm(h, l["ln1_w"], l.get("ln1_b"))
            qkv = R.gemm(xn, l["w_qkv"], l.get("b_qkv"), None, 1 if "b_qkv" in l else 0).to(self.dtype)
            q = qkv[:, :Hq * D].reshape(T, Hq, D)
            k = qkv[:, Hq * D:(Hq + Hkv) * D].reshape(T, Hkv, D)
            v = qkv[:, (Hq + Hkv) * D:].reshape(T, Hkv, D)
            if spec.is_llama:
                q = R.rope(q, self.cos, self.sin, positions).to(self.dtype)
                k = R.rope(k, self.cos, self.sin, positions).to(self.dtype)
            attn = torch.empty(T, Hq * D, dtype=self.dtype, device=self.device)
            t0 = 0
            for slot, qlen, ctx in zip(batch.slots, batch.q_lens, batch.ctx_lens):
                if qlen == 0:
                    continue
                kv = self._slot_cache(slot)[li]
                if kv[0].shape[0] != ctx:
                    # a chunk may be re-sent after a failed step: keep exactly `ctx` cached tokens
                    assert kv[0].shape[0] >= ctx, f"slot {slot}: cache has {kv[0].shape[0]} tokens, step says {ctx}"
                    kv[0], kv[1] = kv[0][:ctx], kv[1][:ctx]
                kv[0] = torch.cat([kv[0], k[t0:t0 + qlen]])
                kv[1] = torch.cat([kv[1], v[t0:t0 + qlen]])
                o = R.attention(q[t0:t0 + qlen], kv[0], kv[1], ctx, scale)
                attn[t0:t0 + qlen] = o.reshape(qlen, Hq * D).to(self.dtype)
                t0 += qlen
            h = R.gemm(attn, l["w_o"], l.get("b_o"), h, 3 if "b_o" in l else 2).to(self.dtype)
          if "w_down" in l:
            xn = self._norm(h, l["ln2_w"], l.get("ln2_b"))
            if spec.is_llama:
                act = R.gemm(xn, torch.cat([l["w_gate"], l["w_up"]], 0), None, None, 4).to(self.dtype)
                h = R.gemm(act, l["w_down"], None, h, 2).to(self.dtype)
            else:
                act = R.gemm(xn, l["w_fc"], l["b_fc"], None, 5).to(self.dtype)
                h = R.gemm(act, l["w_down"], l["b_down"], h, 3).to(self.dtype)
        if not self.plan.has_head:
            return h
        # last stage: only the final position of each sequence that asks for a token
        rows, t0 = [], 0
        for qlen, s in zip(batch.q_lens, batch.sample):
            if s and qlen > 0:
                rows.append(t0 + qlen - 1)
            t0 += qlen
        if not rows:
            return torch.empty(0, spec.vocab_size, dtype=torch.float32, device=self.device)
        hn = self._norm(h[rows], self.w.final_norm_w, self.w.final_norm_b)
        return R.gemm(hn, self.w.lm_head).float()

    # -- sampling -------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, batch: StepBatch, logits: torch.Tensor) -> torch.Tensor:
        """One token per sampled sequence, in batch order. int32 [num_samples]."""
        out, r = [], 0
        for i in range(batch.num_seqs):
            if not batch.sample[i] or batch.q_lens[i] == 0:
                continue
            z = logits[r].float().cpu()
            r += 1
            t = batch.temperature[i]
            if t <= 0.0:
                out.append(int(torch.argmax(z)))
                continue
            probs = filter_logits(z, t, batch.top_k[i], batch.top_p[i])
            g = torch.Generator()
            g.manual_seed((int(batch.seeds[i]) * 1000003 + int(batch.sample_steps[i])) & 0x7FFFFFFFFFFFFFFF)
            out.append(int(torch.multinomial(probs, 1, generator=g)))
        return torch.tensor(out, dtype=torch.int32)
