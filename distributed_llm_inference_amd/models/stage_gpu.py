"""MI355X pipeline-stage executor: a contiguous range of decoder layers (plus embedding on the
first stage and final norm + lm_head + sampler on the last) running entirely on this engine's
HIP kernels, with a paged KV cache for the stage's own layers.

Hot path per decoder layer (Llama):
    norm -> qkv GEMM -> rope + paged-KV write -> paged attention -> o GEMM (+residual)
         -> norm -> gate/up GEMM (silu*mul epilogue) -> down GEMM (+residual)
GEMMs switch between the weight-streaming kernel (<= 64 rows: decode) and the tiled MFMA
kernel (prefill) on the row count. Decode steps of a micro-batch are captured once into a
hipGraph (torch.cuda.CUDAGraph) over static buffers and replayed; the per-step device state
(context lengths, RNG counters) is advanced inside the graph, so a steady-state decode step
needs no host->device traffic at all.

Same interface as `RefStage`; what it replaces in the reference is `Worker.process`
(Worker1.py:82-177) and the orchestrator's embed / norm / lm_head / sampling
(orchestration.py:111,140-169).
"""
from __future__ import annotations

import gc
import math
import os
import threading
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import ops
from ..ops import packing
from ..runtime.batch import StepBatch
from .partition import StagePlan
from .rope import build_rope_table
from .spec import ModelSpec
from .weights import StageWeights

BLOCK = packing.KV_BLOCK
DEFAULT_DECODE_ROWS = 256      # row count up to which GEMMs take the weight-streaming kernel (with split-K partials)


class _PinnedRing:
    """Pinned host staging for step metadata. Everything a step tells the device — token positions,
    sequence slots, sampling parameters, block-table rows — is packed into ONE of these buffers and
    leaves in one asynchronous copy, so building a step never waits for the device (a pageable
    `tensor.to(device)` does: it is ordered behind the stream's queued work and blocks the host until
    it has run). A buffer is reused only after the copy that read it has executed."""

    def __init__(self, words: int, slots: int = 8):
        self.words = words
        self.bufs = [torch.empty(words, dtype=torch.int32).pin_memory() for _ in range(slots)]
        self.views = [b.numpy() for b in self.bufs]
        self.events: List[Optional[torch.cuda.Event]] = [None] * slots
        self.i = 0

    def take(self):
        k = self.i
        self.i = (k + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        return k, self.views[k], self.bufs[k]

    def sent(self, k: int):
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev


@dataclass
class _Work:
    """Activation workspaces of one step in flight. The eager path owns one (sized for a prefill
    chunk); every decode micro-batch owns its own (sized for the micro-batch), so that the graphs
    of different micro-batches can run concurrently on different streams."""
    h: torch.Tensor              # bf16 [T, H]   residual stream
    xn: torch.Tensor             # bf16 [T, H]   normalised input of the next GEMM
    qkv: torch.Tensor            # bf16 [T, (Hq + 2 Hkv) D]
    attn: torch.Tensor           # bf16 [T, Hq D]
    act: torch.Tensor            # bf16 [T, I]
    ws: torch.Tensor             # fp32 split-K partials [S, M, N], reused by every GEMM of the step
    attn_ws: torch.Tensor        # fp32 split-KV partials
    xq: Optional[torch.Tensor]       # uint8 [T, max K] e4m3 activations (fp8 mode), one GEMM input at a time
    xs: Optional[torch.Tensor]       # fp32 [T]         their per-row scales
    head_x: Optional[torch.Tensor]   # bf16 [B, H]      (last stage)
    logits: Optional[torch.Tensor]   # fp32 [B, vocab_pad]
    cmax: Optional[torch.Tensor] = None   # fp32 [B, vocab_pad / 32]: maxima of the logits' 32-column chunks (large-M lm_head)
    guided: bool = False             # the latest lm_head over this workspace left chunk maxima for the sampler


@dataclass
class _MicroBatchState:
    """Static device buffers (and the captured graph) of one in-flight decode micro-batch."""
    B: int
    params: torch.Tensor         # int32 [9 B] the per-row fields below, contiguous: a new composition is ONE upload
    slots: torch.Tensor          # int32 [B]   KV slot of each row (= tok_seq for decode)
    ctx: torch.Tensor            # int32 [B]   cached tokens (= tok_pos for decode); advanced in-graph
    ids: torch.Tensor            # int32 [B]   stage-0 input tokens
    hidden_in: Optional[torch.Tensor]    # bf16 [B, H] (stages > 0)
    hidden_out: Optional[torch.Tensor]   # bf16 [B, H] (stages < last)
    tokens: Optional[torch.Tensor]       # int32 [B]  (last stage)
    temperature: torch.Tensor
    top_k: torch.Tensor
    top_p: torch.Tensor
    seeds: torch.Tensor
    steps: torch.Tensor          # int32 [B] RNG counters; advanced in-graph
    wk: Optional[_Work] = None
    graph: Optional[torch.cuda.CUDAGraph] = None
    splits: int = 1              # split-KV factor the captured step was built with
    splits_wanted: int = 1       # split-KV factor the current composition calls for (re-capture when it differs)


# One capture at a time per process. Entering a capture empties the allocator's cache (`hipFree`, which waits
# for the device) and collects garbage; with several stages in ONE process (in-process pipelines: tests,
# `LocalPipeline`) a second stage thread doing that while the first is mid-capture invalidated the first's
# capture now and then (`hipErrorStreamCaptureInvalidated` at capture end, the stage's engine then fails every
# request). A stage process of a real pipeline holds one stage: the lock is never contended there.
_CAPTURE_LOCK = threading.RLock()


class GpuStage:
    backend = "hip"

    def __init__(self, spec: ModelSpec, plan: StagePlan, weights: StageWeights, *, device="cuda",
                 max_slots: int = 256, max_seq_len: int = 2048, num_blocks: Optional[int] = None,
                 max_tokens: int = 2048, max_batch: int = 64, use_graphs: bool = True,
                 kv_mem_fraction: float = 0.85, weight_dtype: str = "bf16", kv_dtype: str = "bf16",
                 decode_states: int = 1, defer_kv: bool = False, graph_sizes: Optional[List[int]] = None,
                 decode_rows: Optional[int] = None):
        """`num_blocks`: KV blocks of this stage's cache; None sizes it from the device memory that is
        free once weights and workspaces are resident (`kv_mem_fraction` of it, less what the
        `decode_states` micro-batch workspaces will take). `defer_kv`: do not allocate the cache
        yet — the caller reads `kv_budget_blocks()`, agrees on one count with the other stages and
        calls `allocate_kv()`. `decode_rows`: steps of up to this many tokens run the weight-streaming
        GEMMs (default 256, or DLIA_DECODE_ROWS from the environment of the process building the stage)."""
        if not torch.cuda.is_available():
            raise RuntimeError("GpuStage needs a GPU; use RefStage on CPU")
        if spec.head_dim not in (64, 128):
            raise ValueError("the HIP attention kernels support head_dim 64 and 128")
        if spec.num_heads % spec.num_kv_heads or not 1 <= spec.group_size <= 16:
            raise ValueError(f"{spec.name}: {spec.num_heads} query heads over {spec.num_kv_heads} kv heads — the HIP attention "
                             "kernels take GQA groups of 1 to 16 query heads")
        spec.check_context(max_seq_len)
        if weight_dtype not in ("bf16", "fp8", "fp8w"):
            raise ValueError("weight_dtype must be bf16, fp8 or fp8w")
        if kv_dtype not in ("bf16", "fp8"):
            raise ValueError("kv_dtype must be bf16 or fp8")
        self.kv8 = kv_dtype == "fp8"
        self.decode_rows = int(decode_rows if decode_rows is not None else os.environ.get("DLIA_DECODE_ROWS", DEFAULT_DECODE_ROWS))
        self._guided_ok = os.environ.get("DLIA_GUIDED_SAMPLER", "1") != "0"      # A/B switch: 0 = the sampler reads whole rows
        self.K = ops.load()
        # decoder-layer GEMMs: "fp8" = e4m3 weights (per-channel scale) x e4m3 activations (per-row
        # dynamic scale) on the block-scaled fp8 MFMA; "fp8w" = e4m3 weights widened to bf16 in
        # registers, bf16 activations
        self.wmode = weight_dtype
        self.w8 = weight_dtype != "bf16"
        self.spec, self.plan = spec, plan
        self.device = torch.device(device)
        self.max_slots, self.max_seq_len = max_slots, max_seq_len
        self.max_blocks = (max_seq_len + BLOCK - 1) // BLOCK
        self.max_tokens = max(max_tokens, max_batch)
        self.max_batch = max_batch
        # row counts a micro-batch may run its fixed-shape decode step at (ascending, max_batch last)
        self.graph_sizes = sorted({int(b) for b in (graph_sizes or []) if 0 < int(b) < max_batch}) + [max_batch]
        self.use_graphs = use_graphs
        self.null_slot = max_slots            # extra block-table row that never owns blocks
        dev, bf = self.device, torch.bfloat16
        H, I, V = spec.hidden_size, spec.intermediate_size, spec.vocab_size

        # ---- weights: move, cast, pack into MFMA-fragment order ---------------------------
        g = lambda t: None if t is None else t.to(device=dev, dtype=bf).contiguous()
        self.embed = g(weights.embed) if plan.has_embed else None
        self.pos_embed = g(weights.pos_embed) if plan.has_embed else None
        self.layers: List[Dict[str, torch.Tensor]] = []
        for i in range(len(weights.layers)):
            l = weights.layers[i]
            # a stage boundary may fall between a layer's attention and MLP blocks: the first / last
            # layer of a stage then holds the tensors of one block only
            p: Dict[str, torch.Tensor] = {}
            if "w_qkv" in l:
                p["ln1_w"] = g(l["ln1_w"])
                self._pack(p, "w_qkv", g(l["w_qkv"]))
                self._pack(p, "w_o", g(l["w_o"]))
            if "w_down" in l:
                p["ln2_w"] = g(l["ln2_w"])
                self._pack(p, "w_down", g(l["w_down"]))
                if spec.is_llama:
                    self._pack(p, "w_gu", packing.interleave_gate_up(g(l["w_gate"]), g(l["w_up"])))
                else:
                    self._pack(p, "w_fc", g(l["w_fc"]))
            for k in ("ln1_b", "ln2_b", "b_qkv", "b_o", "b_fc", "b_down"):      # GPT-2; q/k/v (and o) of Qwen2 / biased Llama
                if k in l:
                    p[k] = g(l[k])
            self.layers.append(p)
            weights.layers[i] = {}                 # drop the logical copy as we go (peak memory)
        self.vocab_pad = packing.padded_n(V)
        if plan.has_head:
            self.final_norm_w = g(weights.final_norm_w)
            self.final_norm_b = g(weights.final_norm_b)
            self.lm_head = packing.pack_weight(g(weights.lm_head))
        if spec.is_llama:
            self.cos, self.sin = build_rope_table(spec, max_seq_len, dev)
        else:
            self.cos = self.sin = torch.zeros(1, spec.head_dim // 2, device=dev)

        # ---- activation workspaces ----------------------------------------------------------
        self.max_splits = 16
        # split-K partial workspace [S, M, N]: one buffer per step in flight, reused by every GEMM
        # of the step (each partial is consumed by the next kernel before the next GEMM writes)
        need = 1
        for (n_, k_, e_) in ((spec.qkv_size, H, 6), (H, spec.q_size, 6), (H, I, 6),
                             (2 * I if spec.is_llama else I, H, 4 if spec.is_llama else 5),
                             (packing.padded_n(V), H, 0)):
            for m_ in range(1, self.decode_rows + 1):    # any token count up to this takes the weight-streaming path
                sk = int(self.K.gemm_config(m_, n_, k_, e_, True)[1])
                if e_ == 6 or sk > 1:               # direct epilogues of an unsplit GEMM need no partials
                    need = max(need, sk * m_ * n_)
        self._ws_floats = need
        self._gemm_shapes = [(spec.qkv_size, H), (H, spec.q_size), (H, I), (2 * I if spec.is_llama else I, H)]
        self.wk0 = self._new_work(self.max_tokens)
        # step metadata: pinned staging ring (host) and one landing buffer per stream (device)
        self._table_cap = 1 << 16                   # words of one block-table upload; larger updates go in pieces
        self._md_words = 3 * self.max_tokens + 16 * max_batch + 64
        self._ring = _PinnedRing(max(self._md_words, self._table_cap))
        self._landing: Dict[int, torch.Tensor] = {}
        self._eager_sampling = None                 # (step id, device views) staged by `forward` for `sample`
        # ---- paged KV cache for this stage's layers ---------------------------------------
        # (after the workspaces: the cache takes what is left of the device)
        self.kv_mem_fraction, self.decode_states = kv_mem_fraction, max(1, decode_states)
        self.kv = self.kv_scale = None
        self.num_blocks = 0
        self.block_table = torch.full((max_slots + 1, self.max_blocks), -1, device=dev, dtype=torch.int32)
        if not defer_kv:
            self.allocate_kv(num_blocks if num_blocks is not None else self.kv_budget_blocks())
        self._mbs: Dict[tuple, _MicroBatchState] = {}      # (micro-batch, rows) -> static decode state
        self._gemm_cfg: Dict[tuple, tuple] = {}
        self._ones = torch.ones(H, device=dev, dtype=bf)
        self._arange = torch.arange(max_batch + 1, device=dev, dtype=torch.int32)      # q_start of a decode step: one token per row
        self.scale = 1.0 / math.sqrt(spec.head_dim)
        self.graph_replays = 0
        self.graph_captures = 0

    def _new_work(self, rows: int) -> _Work:
        s, dev, bf = self.spec, self.device, torch.bfloat16
        z = lambda *shape, dt=bf: torch.zeros(*shape, device=dev, dtype=dt)
        B = min(rows, self.max_batch)            # sequences (sampled rows) a step over this workspace can have
        # GEMM workspace: split-K partials of the weight-streaming kernels (up to `decode_rows` rows) in
        # front, or the slabs of the large-M kernel's in-launch split-K; the last 4096 words are that
        # kernel's arrival counters and must stay zero between launches (nobody else writes there)
        ws_floats = self._ws_floats
        if self.wmode in ("bf16", "fp8"):        # (the e4m3 kernel splits at most as far as the bf16 one)
            for m_ in {rows, min(rows, 512), min(rows, 1024)}:
                if m_ > self.decode_rows:
                    ws_floats = max([ws_floats] + [int(self.K.gemm_ws_floats(m_, n_, k_)) for n_, k_ in self._gemm_shapes])
        return _Work(
            h=z(rows, s.hidden_size), xn=z(rows, s.hidden_size), qkv=z(rows, s.qkv_size), attn=z(rows, s.q_size),
            act=z(rows, s.intermediate_size), ws=z(ws_floats + 4096, dt=torch.float32),
            attn_ws=z(int(self.K.attention_ws_floats(B, 1, s.num_heads, s.num_kv_heads, s.head_dim, self.max_splits)),
                      dt=torch.float32),
            xq=z(rows, max(s.hidden_size, s.q_size, s.intermediate_size), dt=torch.uint8) if self.wmode == "fp8" else None,
            xs=z(rows, dt=torch.float32) if self.wmode == "fp8" else None,
            head_x=z(B, s.hidden_size) if self.plan.has_head else None,
            logits=z(B, self.vocab_pad, dt=torch.float32) if self.plan.has_head else None,
            cmax=z(B, (self.vocab_pad + 255) // 256 * 8, dt=torch.float32) if self.plan.has_head else None)

    def _pack(self, p: Dict[str, torch.Tensor], name: str, w: torch.Tensor):
        """Packs one GEMM weight into MFMA-fragment order; fp8 mode also stores `<name>_s` scales.
        (The lm_head stays bf16: logits feed a sampler and are the most precision-sensitive GEMM.)"""
        if self.wmode == "fp8":
            p[name], p[name + "_s"] = packing.pack_weight_a8w8(w)
        elif self.wmode == "fp8w":
            p[name], p[name + "_s"] = packing.pack_weight_fp8(w)
        else:
            p[name] = packing.pack_weight(w)

    def _gemm(self, wk, out, x, p, name, bias, residual, ws, N, epi, splitk, ntw):
        sc = p.get(name + "_s")
        if sc is None:
            self.K.gemm(out, x, p[name], bias, residual, ws, N, epi, splitk, ntw)
        elif self.wmode == "fp8w":
            self.K.gemm_w8(out, x, p[name], sc, bias, residual, ws, N, epi, splitk, ntw)
        elif x is None:          # the preceding norm already wrote the e4m3 rows and their scales
            xq = wk.xq[:out.shape[0], :self.spec.hidden_size]
            self.K.gemm_a8w8(out, xq, wk.xs, p[name], sc, bias, residual, ws, N, epi, splitk, min(ntw, 2))
        else:
            T, Kd = x.shape
            xq = wk.xq[:T, :Kd]
            self.K.quant_rows(xq, wk.xs, x)
            self.K.gemm_a8w8(out, xq, wk.xs, p[name], sc, bias, residual, ws, N, epi, splitk, min(ntw, 2))

    def _kvwrite(self, li, qkv, part, S, pbias, tok_pos, tok_seq):
        s = self.spec
        li -= 1 if self.plan.skip_attn_first else 0        # caches are numbered over this stage's attention blocks
        if self.kv8:
            self.K.rope_kvwrite8(qkv, part, S, pbias, self.kv[li], self.kv_scale[li], self.cos, self.sin, tok_pos, tok_seq,
                                 self.block_table, s.num_heads, s.num_kv_heads, s.head_dim, s.is_llama)
        else:
            self.K.rope_kvwrite(qkv, part, S, pbias, self.kv[li], self.cos, self.sin, tok_pos, tok_seq,
                                self.block_table, s.num_heads, s.num_kv_heads, s.head_dim, s.is_llama)

    def _attention(self, li, attn, qkv, q_start, ctx, slots, partial, max_q, splits):
        s = self.spec
        li -= 1 if self.plan.skip_attn_first else 0
        if self.kv8:
            self.K.paged_attention8(attn, qkv, self.kv[li], self.kv_scale[li], q_start, ctx, self.block_table, slots, partial,
                                    max_q, s.num_heads, s.num_kv_heads, s.head_dim, self.scale, splits)
        else:
            self.K.paged_attention(attn, qkv, self.kv[li], q_start, ctx, self.block_table, slots, partial,
                                   max_q, s.num_heads, s.num_kv_heads, s.head_dim, self.scale, splits)

    # ---- KV cache sizing ------------------------------------------------------------------
    def kv_block_bytes(self) -> int:
        """Bytes of one 32-token block over this stage's layers (e4m3: one byte per element plus a
        fp32 scale per token, kv head and K|V)."""
        s = self.spec
        per_elem = (s.head_dim + 4) if self.kv8 else s.head_dim * 2
        return max(1, len(self.plan.attn_layers)) * s.num_kv_heads * 2 * BLOCK * per_elem

    def _decode_state_bytes(self) -> int:
        """What one decode micro-batch will allocate on first use (activations, split-K partials,
        logits), so the cache does not take memory the serving loop still needs."""
        s = self.spec
        per_tok = 2 * (2 * s.hidden_size + s.qkv_size + s.q_size + s.intermediate_size)
        n = 0
        for B in self.graph_sizes:               # one state per row count a micro-batch may run at
            n += B * per_tok + 4 * self._ws_floats + 4 * self.wk0.attn_ws.numel() * B // self.max_batch
            if self.plan.has_head:
                n += B * (4 * self.vocab_pad + 2 * s.hidden_size)
            if self.wmode == "fp8":
                n += B * max(s.hidden_size, s.q_size, s.intermediate_size)
        return n

    def kv_budget_blocks(self, share: float = 1.0) -> int:
        """Blocks the cache may hold: `kv_mem_fraction` of the memory free right now (weights and
        the eager workspace are resident), times `share` when several stages split one device, less
        the decode micro-batch workspaces and 2 GiB for communicators, graphs and fragmentation;
        never more than every slot at full length could use."""
        torch.cuda.empty_cache()               # blocks cached by the weight packing count as free
        free, _ = torch.cuda.mem_get_info(self.device)
        budget = int(free * self.kv_mem_fraction * share) - self.decode_states * self._decode_state_bytes() - (2 << 30)
        blocks = budget // self.kv_block_bytes()
        return int(max(16, min(blocks, self.max_slots * self.max_blocks)))

    def allocate_kv(self, num_blocks: int):
        s, L, dev = self.spec, len(self.plan.attn_layers), self.device      # one cache per attention block of this stage
        self.num_blocks = int(num_blocks)
        # zero-filled: masked-out (never written) slots must hold finite values, P=0 times NaN is NaN
        self.kv = torch.zeros(max(L, 1), self.num_blocks, s.num_kv_heads, 2, BLOCK * s.head_dim,
                              device=dev, dtype=torch.uint8 if self.kv8 else torch.bfloat16)
        self.kv_scale = (torch.zeros(max(L, 1), self.num_blocks, s.num_kv_heads, 2, BLOCK, device=dev)
                         if self.kv8 else None)

    # ---- step metadata: host -> device ------------------------------------------------------
    def _land(self) -> torch.Tensor:
        """The current stream's landing buffer: uploads and the kernels that read them are ordered
        by the stream, so one buffer per stream is enough (eager steps run on the main stream,
        every decode micro-batch on its own)."""
        key = torch.cuda.current_stream(self.device).cuda_stream
        buf = self._landing.get(key)
        if buf is None:
            buf = torch.zeros(self._ring.words, device=self.device, dtype=torch.int32)
            self._landing[key] = buf
        return buf

    def _upload(self, parts, dst: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
        """Packs 1-D numpy arrays (int32 / float32 / int64) into one pinned buffer and copies it to
        `dst` (default: the stream's landing buffer) asynchronously. Returns typed device views."""
        dst = self._land() if dst is None else dst
        k, hv, hb = self._ring.take()
        off, spans = 0, []
        for a in parts:
            a = np.ascontiguousarray(a)
            if a.dtype == np.int64:
                off += off & 1                      # 8-byte alignment of int64 views
            w = a.size * (a.dtype.itemsize // 4)
            if off + w > self._ring.words or off + w > dst.numel():
                raise ValueError(f"step metadata of {off + w} words exceeds the staging buffer ({min(self._ring.words, dst.numel())})")
            hv[off:off + w] = a.view(np.int32)
            spans.append((off, w, a.dtype))
            off += w
        if off:
            dst[:off].copy_(hb[:off], non_blocking=True)
        self._ring.sent(k)
        tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.int64): torch.int64}
        return [dst[o:o + w] if dt not in tdt else dst[o:o + w].view(tdt[dt]) for o, w, dt in spans]

    # ---- block tables ---------------------------------------------------------------------
    def _write_table_rows(self, slots: List[int], tables: List[List[int]]):
        """block_table[slot] = blocks (padded with -1) for each pair: rows travel through the staging
        ring and a scatter kernel, `_table_cap` words at a time."""
        n, mb = len(slots), self.max_blocks
        if n == 0:
            return
        vals = np.full((n, mb), -1, dtype=np.int32)
        for i, (slot, blocks) in enumerate(zip(slots, tables)):
            if not (0 <= slot < self.max_slots):
                raise ValueError(f"slot {slot} out of range")
            if len(blocks) > mb:
                raise ValueError(f"bad block table for slot {slot}")
            if blocks:
                vals[i, :len(blocks)] = blocks
                if min(blocks) < 0 or max(blocks) >= self.num_blocks:
                    raise ValueError(f"bad block table for slot {slot}")
        rows = np.asarray(slots, dtype=np.int32)
        per = max(1, self._table_cap // (mb + 1))
        for i in range(0, n, per):
            r, v = self._upload([rows[i:i + per], vals[i:i + per].reshape(-1)])
            self.K.scatter_rows(self.block_table, r, v.view(-1, mb))

    def set_tables(self, batch: StepBatch):
        self._write_table_rows(batch.table_slots, batch.table_blocks)

    def apply_growth(self, batch: StepBatch):
        """block_table[slot][pos] = block for the sequences that got one more block this step.
        Enqueued on the current stream, i.e. ahead of the step (graph replay included)."""
        n = len(batch.grow_slots)
        if n == 0:
            return
        sl, ps, bl = (np.asarray(v, dtype=np.int64) for v in (batch.grow_slots, batch.grow_pos, batch.grow_blocks))
        if ((sl < 0) | (sl >= self.max_slots) | (ps < 0) | (ps >= self.max_blocks) | (bl < 0) | (bl >= self.num_blocks)).any():
            raise ValueError("bad block-table growth entry")
        idx, val = self._upload([(sl * self.max_blocks + ps).astype(np.int32), bl.astype(np.int32)])
        self.K.scatter(self.block_table, idx, val)

    def release(self, slots: List[int]):
        ok = [s for s in slots if 0 <= s < self.max_slots]
        self._write_table_rows(ok, [[]] * len(ok))

    # ---- the layer stack on T tokens ------------------------------------------------------
    def _cfg(self, M: int, N: int, K: int, epi: int):
        key = (M, N, K, epi)
        c = self._gemm_cfg.get(key)
        if c is None:
            c = tuple(int(v) for v in self.K.gemm_config(M, N, K, epi, True))
            self._gemm_cfg[key] = c
        return c

    def _layers(self, wk: _Work, T: int, B: int, tok_pos, tok_seq, q_start, ctx, slots, max_q: int, splits: int,
                alone: bool = False, lead_decode: int = 0):
        """Runs this stage's decoder layers on wk.h[:T]. Returns the pending split-K partial
        (splits, bias) of the last down projection that the caller still has to fold into h
        (decode regime), or None when h is final. `lead_decode`: the first that many sequences of a
        ragged step are single decode tokens riding with prompt chunks; their attention runs on the
        decode-shaped kernel, the chunks' on the prefill-shaped one."""
        K, s = self.K, self.spec
        h, xn, qkv, attn, act = wk.h[:T], wk.xn[:T], wk.qkv[:T], wk.attn[:T], wk.act[:T]
        Hq, Hkv, D, H, I = s.num_heads, s.num_kv_heads, s.head_dim, s.hidden_size, s.intermediate_size
        partial = wk.attn_ws if splits > 1 else None
        llama = s.is_llama
        nmode = 0 if llama else 1
        fp8 = self.wmode == "fp8"

        def attend(li):
            if 0 < lead_decode < B and max_q > 1:
                n = lead_decode
                self._attention(li, attn, qkv, q_start[:n + 1], ctx[:n], slots[:n], partial, 1, splits)
                self._attention(li, attn, qkv, q_start[n:], ctx[n:], slots[n:], partial, max_q, splits)
            else:
                self._attention(li, attn, qkv, q_start, ctx, slots, partial, max_q, splits)

        def norm(part, S, pbias, w, b):
            """Pre-GEMM norm. In fp8 mode its output is already the GEMM's e4m3 input (returns None)."""
            if fp8:
                K.norm_q(wk.xq[:T, :H], wk.xs, h, None, part, S, pbias, w, b, s.norm_eps, nmode)
                return None
            K.norm(xn, h, None, part, S, pbias, w, b, None, s.norm_eps, nmode)
            return xn

        if T > self.decode_rows:
            # compute-bound regime (large decode micro-batches, prefill chunks): tiled MFMA GEMMs with
            # fused bias / residual / activation epilogues; the 256 x 256 kernel (bf16 and e4m3) splits
            # K inside the launch on shapes with fewer tiles than CUs and needs the workspace for that
            lws = wk.ws if self.wmode in ("bf16", "fp8") else None
            # workgroups a thin GEMM is spread over: the whole chip for a step that runs alone (prefill
            # chunk on the eager path), half of it for decode micro-batches that overlap on two streams
            tg = 256 if alone else 1
            for li, p in enumerate(self.layers):
                if "w_qkv" in p:            # attention block (absent from a first layer whose attention ran upstream)
                    x1 = norm(None, 1, None, p["ln1_w"], p.get("ln1_b"))
                    self._gemm(wk, qkv, x1, p, "w_qkv", p.get("b_qkv"), None, lws, s.qkv_size, 1 if "b_qkv" in p else 0, tg, 0)
                    self._kvwrite(li, qkv, None, 1, None, tok_pos, tok_seq)
                    attend(li)
                    self._gemm(wk, h, attn, p, "w_o", p.get("b_o"), h, lws, H, 3 if "b_o" in p else 2, tg, 0)
                if "w_down" not in p:       # MLP block runs on the next stage
                    continue
                x2 = norm(None, 1, None, p["ln2_w"], p.get("ln2_b"))
                if llama:
                    self._gemm(wk, act, x2, p, "w_gu", None, None, lws, 2 * I, 4, tg, 0)
                    self._gemm(wk, h, act, p, "w_down", None, h, lws, H, 2, tg, 0)
                else:
                    self._gemm(wk, act, x2, p, "w_fc", p["b_fc"], None, lws, I, 5, tg, 0)
                    self._gemm(wk, h, act, p, "w_down", p["b_down"], h, lws, H, 3, tg, 0)
            return None
        # decode regime: weight-streaming GEMMs write fp32 split-K partials which the NEXT kernel
        # (norm / rope) sums while it reads them anyway — no reduce launches, no bf16 round trip
        ws = wk.ws
        nq, sq = self._cfg(T, s.qkv_size, H, 6)
        no, so = self._cfg(T, H, s.q_size, 6)
        nd, sd = self._cfg(T, H, I, 6)
        pend = None
        for li, p in enumerate(self.layers):
            if "w_qkv" in p:
                if pend is None:
                    x1 = norm(None, 1, None, p["ln1_w"], p.get("ln1_b"))
                else:
                    x1 = norm(ws, pend[0], pend[1], p["ln1_w"], p.get("ln1_b"))
                self._gemm(wk, qkv, x1, p, "w_qkv", None, None, ws, s.qkv_size, 6, sq, nq)
                self._kvwrite(li, qkv, ws, sq, p.get("b_qkv"), tok_pos, tok_seq)
                attend(li)
                self._gemm(wk, h, attn, p, "w_o", None, None, ws, H, 6, so, no)
                pend = (so, p.get("b_o"))
            if "w_down" not in p:           # MLP block runs on the next stage: the O partials are folded by the caller
                continue
            x2 = norm(None, 1, None, p["ln2_w"], p.get("ln2_b")) if pend is None else \
                norm(ws, pend[0], pend[1], p["ln2_w"], p.get("ln2_b"))
            if llama:
                self._gemm(wk, act, x2, p, "w_gu", None, None, ws, 2 * I, 4, 0, 0)
            else:
                self._gemm(wk, act, x2, p, "w_fc", p["b_fc"], None, ws, I, 5, 0, 0)
            self._gemm(wk, h, act, p, "w_down", None, None, ws, H, 6, sd, nd)
            pend = (sd, p.get("b_down"))
        return pend

    def _fold(self, wk: _Work, T: int, pend):
        """h[:T] += pending partials (stage output that is sent on rather than normalised here)."""
        if pend is not None:
            self.K.norm(wk.xn[:T], wk.h[:T], None, wk.ws, pend[0], pend[1], self._ones, None, None,
                        self.spec.norm_eps, 2)

    def _head(self, wk: _Work, rows_idx: torch.Tensor, R: int, T: int, pend):
        """Final norm on the selected rows of wk.h (folding the pending partials of those rows
        first), then lm_head -> wk.logits[:R]."""
        s, K = self.spec, self.K
        hx = wk.head_x[:R]
        nmode = 0 if s.is_llama else 1
        if pend is None:
            K.norm(hx, wk.h[:T], None, None, 1, None, self.final_norm_w, self.final_norm_b, rows_idx[:R],
                   s.norm_eps, nmode)
        else:
            K.norm(hx, wk.h[:T], None, wk.ws, pend[0], pend[1], self.final_norm_w, self.final_norm_b,
                   rows_idx[:R], s.norm_eps, nmode)
        # The lm_head GEMM (tile kernel above 256 rows, weight-streaming kernels up to there) also leaves the maximum
        # of every 32-column chunk of every row, and the sampler reads the logits only where a top-k candidate can be
        # (a few KB of a 513 KB row).
        wk.guided = wk.cmax is not None and self.vocab_pad % 4 == 0 and s.vocab_size <= 262144 and self._guided_ok
        if wk.guided:
            K.gemm_head(wk.logits[:R], wk.cmax[:R], hx, self.lm_head, wk.ws, self.vocab_pad, s.vocab_size)
        else:
            K.gemm(wk.logits[:R], hx, self.lm_head, None, None, wk.ws, self.vocab_pad, 0, 0, 0)
        return wk.logits[:R]

    # ---- generic (eager) forward: prefill chunks, and decode when graphs are off ---------
    @torch.no_grad()
    def forward(self, batch: StepBatch, x: Optional[torch.Tensor]) -> torch.Tensor:
        s = self.spec
        B, T = batch.num_seqs, batch.num_tokens
        if B > self.max_batch or T > self.max_tokens:
            raise ValueError(f"step with {B} sequences / {T} tokens exceeds the stage limits "
                             f"({self.max_batch} / {self.max_tokens})")
        self.set_tables(batch)
        self.apply_growth(batch)
        wk = self.wk0
        q = np.asarray(batch.q_lens, dtype=np.int64)
        c = np.asarray(batch.ctx_lens, dtype=np.int64)
        slots = np.asarray(batch.slots, dtype=np.int32)
        if B and int((c + q).max()) > self.max_seq_len:
            i = int(np.argmax(c + q))
            raise ValueError(f"sequence in slot {batch.slots[i]} would reach {int(c[i] + q[i])} tokens > max_seq_len {self.max_seq_len}")
        q_start = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(q, out=q_start[1:])
        tok_seq = np.repeat(slots, q)
        tok_pos = (np.arange(T, dtype=np.int64) - np.repeat(q_start[:-1] - c, q)).astype(np.int32)
        sampled = (np.asarray(batch.sample, dtype=np.int64) != 0) & (q > 0)
        rows = (q_start[1:][sampled] - 1).astype(np.int32)
        R = int(rows.size)
        if R > self.max_batch:
            raise ValueError(f"{R} sampled rows exceed max_batch {self.max_batch}")
        dev = self.device
        parts = [tok_pos, tok_seq, q_start.astype(np.int32), c.astype(np.int32), slots, rows]
        host_ids = self.plan.has_embed and not (torch.is_tensor(x) and x.is_cuda)
        if host_ids:
            ids_h = x.numpy() if torch.is_tensor(x) else np.asarray(x)
            ids_h = ids_h.astype(np.int32, copy=False).reshape(-1)
            if ids_h.size != T:
                raise ValueError(f"expected {T} token ids, got {ids_h.size}")
            parts.append(ids_h)
        if self.plan.has_head and R:
            pick = lambda vals, dt: np.asarray(vals, dtype=dt)[sampled]
            parts += [pick(batch.temperature, np.float32), pick(batch.top_k, np.int32), pick(batch.top_p, np.float32),
                      pick(batch.sample_steps, np.int32), pick(batch.seeds, np.int64)]
        up = self._upload(parts)
        md_tok_pos, md_tok_seq, md_q_start, md_ctx, md_slots, md_rows = up[:6]
        nxt = 6
        if self.plan.has_embed:
            if host_ids:
                ids = up[nxt]
                nxt += 1
            else:
                ids = x.to(dtype=torch.int32).reshape(-1)
                if ids.numel() != T:
                    raise ValueError(f"expected {T} token ids, got {ids.numel()}")
            self.K.embed(wk.h[:T], ids, md_tok_pos, self.embed, self.pos_embed)
        else:
            if tuple(x.shape) != (T, s.hidden_size):
                raise ValueError(f"expected hidden [{T},{s.hidden_size}], got {tuple(x.shape)}")
            wk.h[:T].copy_(x.to(device=dev, dtype=torch.bfloat16), non_blocking=True)
        self._eager_sampling = (batch.step_id, R, up[nxt:nxt + 5]) if (self.plan.has_head and R) else None
        max_q = int(q.max()) if B else 1
        lead = 0                                   # decode tokens riding in front of prompt chunks (Engine._submit_prefill)
        if max_q > 1:
            ones = np.flatnonzero(q != 1)
            lead = int(ones[0]) if ones.size else 0
        pend = self._layers(wk, T, B, md_tok_pos, md_tok_seq, md_q_start, md_ctx, md_slots, max_q, 1, alone=True,
                            lead_decode=lead)
        if not self.plan.has_head:
            self._fold(wk, T, pend)
            return wk.h[:T]
        if R == 0:
            return wk.logits[:0, :s.vocab_size]
        return self._head(wk, md_rows, R, T, pend)[:, :s.vocab_size]

    @torch.no_grad()
    def sample(self, batch: StepBatch, logits: torch.Tensor) -> torch.Tensor:
        """Tokens (device int32 [num_samples]) for the rows `forward` returned logits for."""
        idx = [i for i in range(batch.num_seqs) if batch.sample[i] and batch.q_lens[i] > 0]
        R = len(idx)
        out = torch.empty(R, device=self.device, dtype=torch.int32)
        if R == 0:
            return out
        staged = self._eager_sampling
        if staged is not None and staged[0] == batch.step_id and staged[1] == R:
            temp, top_k, top_p, steps, seeds = staged[2]          # uploaded with the step's other metadata
        else:
            pick = lambda vals, dt: np.asarray([vals[i] for i in idx], dtype=dt)
            temp, top_k, top_p, steps, seeds = self._upload([pick(batch.temperature, np.float32), pick(batch.top_k, np.int32),
                                                              pick(batch.top_p, np.float32), pick(batch.sample_steps, np.int32),
                                                              pick(batch.seeds, np.int64)])
        lg = self.wk0.logits[:R]  # `logits` is a view of this buffer (padded columns ignored by `vocab`)
        if self.wk0.guided:
            self.K.sample_guided(out, lg, self.wk0.cmax[:R], temp, top_k, top_p, seeds, steps, self.spec.vocab_size, None, None)
        else:
            self.K.sample(out, lg, temp, top_k, top_p, seeds, steps, self.spec.vocab_size, None, None)
        return out

    # ---- graph-captured decode ------------------------------------------------------------
    def _mb_state(self, mb: int, B: int) -> _MicroBatchState:
        st = self._mbs.get((mb, B))
        if st is not None:
            return st
        # (several stages as threads of one process: a gigabyte of workspace is not allocated, nor an old
        # state with its graph dropped, while another stage's thread is capturing)
        with _CAPTURE_LOCK:
            return self._new_mb_state(mb, B)

    def _new_mb_state(self, mb: int, B: int) -> _MicroBatchState:
        if B not in self.graph_sizes:                      # an odd size (a test, a probe): it replaces the previous odd one
            for k in [k for k in self._mbs if k[0] == mb and k[1] not in self.graph_sizes]:
                del self._mbs[k]
        dev, H = self.device, self.spec.hidden_size
        # per-row fields in one buffer, in upload order: slots | ctx | top_k | steps | temperature | top_p | seeds (int64) | ids
        params = torch.zeros(9 * B + 2, device=dev, dtype=torch.int32)
        field = lambda i, n=1: params[i * B:(i + n) * B]
        params[:B] = self.null_slot
        field(5).view(torch.float32).fill_(1.0)
        st = _MicroBatchState(
            B=B, params=params, slots=field(0), ctx=field(1), ids=field(8),
            hidden_in=None if self.plan.has_embed else torch.zeros(B, H, device=dev, dtype=torch.bfloat16),
            hidden_out=None if self.plan.has_head else torch.zeros(B, H, device=dev, dtype=torch.bfloat16),
            tokens=torch.zeros(B, device=dev, dtype=torch.int32) if self.plan.has_head else None,
            temperature=field(4).view(torch.float32), top_k=field(2), top_p=field(5).view(torch.float32),
            seeds=field(6, 2).view(torch.int64), steps=field(3), wk=self._new_work(B))
        self._mbs[(mb, B)] = st
        return st

    def _decode_body(self, st: _MicroBatchState):
        """One decode step over the micro-batch's static buffers (captured into the graph)."""
        K, B, wk = self.K, st.B, st.wk
        if self.plan.has_embed:
            K.embed(wk.h[:B], st.ids, st.ctx, self.embed, self.pos_embed)
        else:
            wk.h[:B].copy_(st.hidden_in)
        pend = self._layers(wk, B, B, st.ctx, st.slots, self._arange[:B + 1], st.ctx, st.slots, 1, st.splits)
        if self.plan.has_head:
            self._head(wk, self._arange, B, B, pend)
            # the thread that writes a row's token also advances the row's draw counter and context length and — single-stage
            # pipeline, where the feedback hop is local — puts the token into the next step's input ids
            K.sample_step(st.tokens, wk.logits[:B], wk.cmax[:B] if wk.guided else None, st.temperature, st.top_k, st.top_p,
                          st.seeds, st.steps, self.spec.vocab_size, st.ids if self.plan.has_embed else None, st.ctx)
        else:
            self._fold(wk, B, pend)
            st.hidden_out.copy_(wk.h[:B])
            K.advance(st.ctx, 1)

    @torch.no_grad()
    def decode_step(self, batch: StepBatch, x: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Decode step of micro-batch `batch.micro_batch` over static buffers.

        Returns the static output buffer: hidden [B,H] (not the last stage) or sampled tokens
        int32 [B] (last stage). When `batch.replay` is set and nothing about the micro-batch
        changed, this is a bare graph replay. `x` (tokens for stage 0 / hidden otherwise) is
        copied into the static input when given; a caller that received straight into
        `input_buffer()` passes None.
        """
        B = batch.num_seqs
        if B > self.max_batch:
            raise ValueError(f"micro-batch of {B} exceeds max_batch {self.max_batch}")
        st = self._mb_state(batch.micro_batch, B)
        self.apply_growth(batch)
        if not batch.replay:
            self.set_tables(batch)
            ctx = np.asarray(batch.ctx_lens, dtype=np.int32)
            if B and int(ctx.max()) + 1 > self.max_seq_len:
                raise ValueError("sequence exceeds max_seq_len")
            # rows whose q_len is 0 are padding: park them on the null slot
            slots = np.where(np.asarray(batch.q_lens) == 1, np.asarray(batch.slots), self.null_slot).astype(np.int32)
            parts = [slots, ctx, np.asarray(batch.top_k, dtype=np.int32), np.asarray(batch.sample_steps, dtype=np.int32),
                     np.asarray(batch.temperature, dtype=np.float32), np.asarray(batch.top_p, dtype=np.float32),
                     np.asarray(batch.seeds, dtype=np.int64)]
            if self.plan.has_embed and batch.token_ids is not None:
                parts.append(np.asarray(batch.token_ids, dtype=np.int32))
            if any(p_.size != B for p_ in parts):
                raise ValueError("a fixed-shape decode step needs every per-row field for every row")
            self._upload(parts, st.params)          # one asynchronous copy straight into the graph's static fields
            max_ctx = max(batch.ctx_lens) if batch.ctx_lens else 0
            st.splits_wanted = self._pick_splits(B, max_ctx)
        if x is not None:
            (st.ids if self.plan.has_embed else st.hidden_in).copy_(x)
        if not self.use_graphs:
            st.splits = st.splits_wanted
            self._decode_body(st)
        else:
            if st.graph is None or st.splits != st.splits_wanted:
                st.splits = st.splits_wanted
                self._capture(st)
            st.graph.replay()
            self.graph_replays += 1
        return st.tokens if self.plan.has_head else st.hidden_out

    def warm_decode_states(self):
        """With more than one decode step size (`decode_buckets`): allocate every micro-batch's static
        buffers and capture its step at every size now, on padding rows, so that the first request to
        run at a size — the first burst after a quiet hour — does not pay for a gigabyte of workspace
        and a graph capture (measured: 2.6 s of a 512-client burst). One size only: left to first use."""
        if not self.use_graphs or len(self.graph_sizes) < 2 or self.kv is None:
            return
        for mb in range(self.decode_states):
            for B in self.graph_sizes:
                st = self._mb_state(mb, B)
                if st.graph is None:
                    st.splits = st.splits_wanted = 1
                    self._capture(st)
        torch.cuda.synchronize(self.device)

    def input_buffer(self, mb: int, B: int) -> torch.Tensor:
        st = self._mb_state(mb, B)
        return st.ids if self.plan.has_embed else st.hidden_in

    def _pick_splits(self, B: int, max_ctx: int) -> int:
        """Inter-workgroup split-KV factor. A (sequence, kv head) item already spreads its KV
        blocks over the 4 waves of its workgroup; more splits are only worth their merge kernel
        when there are too few items to fill the chip and the context is long."""
        waves = B * self.spec.num_kv_heads * 4
        if max_ctx < 1024 or waves >= 2048:
            return 1
        s = 1
        while waves * s < 2048 and s < self.max_splits and (max_ctx // (s * 2)) >= 256:
            s *= 2
        return s

    def _capture(self, st: _MicroBatchState):
        # Warm up on a side stream (first-use initialisation must not happen under capture), then
        # capture. The captured step advances ctx/steps (and, on a single-stage pipeline, feeds its
        # tokens back into ids), so save and restore them around both.
        with _CAPTURE_LOCK:
            saved = (st.ctx.clone(), st.steps.clone(), st.ids.clone())
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                self._decode_body(st)
            torch.cuda.current_stream(self.device).wait_stream(side)
            st.ctx.copy_(saved[0]); st.steps.copy_(saved[1]); st.ids.copy_(saved[2])
            g = torch.cuda.CUDAGraph()
            # Thread-local capture mode: only this thread issues the captured work, but a stage process
            # of a multi-GPU pipeline also runs RCCL's watchdog thread (event queries at any time), and
            # under the default global mode a HIP call from another thread invalidates the capture.
            # No cyclic garbage collection inside the capture: the collector runs in whichever thread trips its
            # counter, and what it finds may be another thread's (a closed pipeline's graphs, pools, pinned buffers),
            # whose release is not a thing to do on a capturing thread (seen once: the process aborted inside a
            # collection that started in a capturing stage thread of a 5-stage in-process pipeline).
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._decode_body(st)
            finally:
                if gc_was_on:
                    gc.enable()
            st.ctx.copy_(saved[0]); st.steps.copy_(saved[1]); st.ids.copy_(saved[2])
            st.graph = g
            self.graph_captures += 1

    def drop_graphs(self):
        """Drops the captured graphs (with their memory pools) and the static decode buffers now, on the calling
        thread. A pipeline that is closed does this instead of leaving it to whenever the last reference goes:
        were that the cyclic collector, it would run on whichever thread trips it — possibly a stage thread of
        the next pipeline, in the middle of its steps or of a capture."""
        with _CAPTURE_LOCK:
            torch.cuda.synchronize(self.device)          # no replay of theirs is still running
            for st in self._mbs.values():
                st.graph = None
            self._mbs.clear()

    # ---- introspection ----------------------------------------------------------------------
    def memory_report(self) -> Dict[str, float]:
        gib = 1 << 30
        wbytes = sum(t.numel() * t.element_size() for l in self.layers for t in l.values())
        for t in (self.embed, self.pos_embed, getattr(self, "lm_head", None)):
            if t is not None:
                wbytes += t.numel() * t.element_size()
        return {"weights_gib": wbytes / gib, "kv_gib": (self.kv.numel() * self.kv.element_size() + (self.kv_scale.numel() * 4 if self.kv8 else 0)) / gib, "kv_blocks": self.num_blocks,
                "kv_tokens": self.num_blocks * BLOCK}
