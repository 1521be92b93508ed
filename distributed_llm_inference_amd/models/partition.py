"""Split a model's decoder layers into contiguous pipeline stages.

The reference hard-wires two workers with hand-edited half-open ranges
(`LAYER_START/LAYER_END`, Worker1.py:27-28 = [0,11), Worker2.py:26-27 = [11,22)) and keeps
embedding, final norm and lm_head on the orchestrator (orchestration.py:45-47).  Here the
first stage owns the embedding, the last stage owns final norm + lm_head + sampler, and the
layer ranges are either given explicitly, split evenly (which reproduces 11/11 for 22
layers), or balanced by cost so that the lm_head (2.4 decoder layers' worth of weights and FLOPs
for Llama-3-8B, about 1.5 layers' worth of time) does not make the last stage the bottleneck.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

from .spec import ModelSpec


@dataclass(frozen=True)
class StagePlan:
    stage: int
    num_stages: int
    layer_start: int          # half-open [layer_start, layer_end), as in the reference
    layer_end: int
    has_embed: bool
    has_head: bool
    # A decoder layer is two blocks, attention (norm, QKV, RoPE + KV write, attention, O + residual) and
    # MLP (norm, gate/up, down + residual), and only the hidden state crosses between them — the same
    # [tokens, hidden] rows that cross between layers. A stage boundary may therefore fall INSIDE a
    # layer: that layer then appears in two adjacent plans, attention block on the first
    # (`attn_only_last`), MLP block on the second (`skip_attn_first`). With whole layers 8 stages
    # of Llama-3-8B are 4,4,4,4,4,5,5,2 + head and the slowest stage bounds weak scaling at 0.84;
    # with half layers the bound is 0.95.
    skip_attn_first: bool = False
    attn_only_last: bool = False

    @property
    def num_layers(self) -> int:
        return self.layer_end - self.layer_start

    def has_attn(self, layer: int) -> bool:
        return self.layer_start <= layer < self.layer_end and not (self.skip_attn_first and layer == self.layer_start)

    def has_mlp(self, layer: int) -> bool:
        return self.layer_start <= layer < self.layer_end and not (self.attn_only_last and layer == self.layer_end - 1)

    @property
    def attn_layers(self) -> List[int]:
        return [i for i in self.layers if self.has_attn(i)]

    @property
    def mlp_layers(self) -> List[int]:
        return [i for i in self.layers if self.has_mlp(i)]

    @property
    def layers(self) -> range:
        return range(self.layer_start, self.layer_end)

    def describe(self) -> str:
        # same "start-end" spelling the reference's worker /health uses (Worker1.py:204); x.5 = the
        # boundary between a layer's attention and MLP blocks
        a = f"{self.layer_start}.5" if self.skip_attn_first else f"{self.layer_start}"
        b = f"{self.layer_end - 1}.5" if self.attn_only_last else f"{self.layer_end}"
        return f"{a}-{b}"


def even_split(num_layers: int, num_stages: int) -> List[Tuple[int, int]]:
    """Contiguous ranges whose sizes differ by at most one; earlier stages get the extras."""
    if num_stages < 1:
        raise ValueError("num_stages must be >= 1")
    if num_stages > num_layers:
        raise ValueError(f"cannot split {num_layers} layers into {num_stages} stages")
    base, extra = divmod(num_layers, num_stages)
    out, start = [], 0
    for s in range(num_stages):
        n = base + (1 if s < extra else 0)
        out.append((start, start + n))
        start += n
    return out


# Time of final norm + lm_head + sampler relative to their share of parameters. Measured on MI355X
# in the regime every stage of a full node runs in (bench/stage_cost.py: Llama-3-8B, micro-batches of
# 2048 rows alternating over two HIP streams, hipGraph decode, ~150 cached tokens): a step of two
# micro-batches costs 1.842 ms per decoder layer (4 and 8 layers: 7.37 / 14.74 ms) and 4.07 ms for the
# head (2 or 3 layers + head: 7.76 / 9.60 ms) = 2.21 layers for 2.41 layers' worth of parameters.
# (Single-stream kernel times say 1.5-1.7 layers: the lm_head fills every CU while a layer's thin
# GEMMs leave CUs to the other stream, so the head weighs more once two streams share the chip. In
# the weight-streaming regime of small batches a block costs what its parameters weigh: ~1.)
HEAD_TIME_PER_PARAM = 0.92


# Share of a decoder layer's time spent in its attention block (the rest is the MLP block) at large
# micro-batches, same measurement: 4 layers + one attention block 8.25 ms, one more MLP block 0.94 ms
# on top of 4 layers = 0.88 : 0.96. By parameters (what a small, weight-streaming batch pays for) the
# attention block of Llama-3-8B is 19 %.
ATTN_BLOCK_SHARE = 0.48
COMPUTE_BOUND_ROWS, WEIGHT_BOUND_ROWS = 512, 128


def attn_block_share(spec: ModelSpec, rows: Optional[int] = None) -> float:
    """Attention block's share of a layer's time for micro-batches of `rows` sequences: the measured
    compute-bound share from 512 rows up (and when `rows` is not given), the parameter share up to
    128 rows, linear in between."""
    if rows is None or rows >= COMPUTE_BOUND_ROWS:
        return ATTN_BLOCK_SHARE
    by_params = spec.attn_param_count() / spec.layer_param_count()
    if rows <= WEIGHT_BOUND_ROWS:
        return by_params
    f = (rows - WEIGHT_BOUND_ROWS) / (COMPUTE_BOUND_ROWS - WEIGHT_BOUND_ROWS)
    return by_params + f * (ATTN_BLOCK_SHARE - by_params)


def head_cost_layers(spec: ModelSpec) -> float:
    """What the last stage's head costs, in decoder layers."""
    return HEAD_TIME_PER_PARAM * spec.head_param_count() / spec.layer_param_count()


def stage_costs(spec: ModelSpec, plans: Sequence["StagePlan"], rows: Optional[int] = None) -> List[float]:
    """Estimated cost of every stage in decoder layers (its layers, plus the head on the last)."""
    a = attn_block_share(spec, rows)
    return [a * len(p.attn_layers) + (1.0 - a) * len(p.mlp_layers)
            + (head_cost_layers(spec) if p.has_head else 0.0) for p in plans]


def weak_scaling_bound(spec: ModelSpec, plans: Sequence["StagePlan"], rows: Optional[int] = None) -> float:
    """Upper bound on pipeline efficiency set by the partition alone: every stage waits for the
    slowest one, so N stages deliver at most mean cost / max cost of N perfectly balanced ones
    (Llama-3-8B over 8 stages of whole layers: 4,4,4,5,5,4,4,2 + head -> 34.2 / (8 x 5) = 0.86)."""
    c = stage_costs(spec, plans, rows)
    return sum(c) / (len(c) * max(c))


def balanced_split(spec: ModelSpec, num_stages: int) -> List[Tuple[int, int]]:
    """Minimise the heaviest stage.

    Cost unit = one decoder layer.  The last stage additionally runs final norm + lm_head +
    sampler (`head_cost_layers`); the embedding is a row gather and costs ~nothing.  Exact DP
    over contiguous splits (L ≤ a few hundred, P ≤ 8: trivial).
    """
    L = spec.num_layers
    if num_stages == 1:
        return [(0, L)]
    if num_stages > L:
        raise ValueError(f"cannot split {L} layers into {num_stages} stages")
    head_cost = head_cost_layers(spec)
    INF = float("inf")
    # best[s][i]: minimal max-cost of splitting layers [0,i) into s stages (all non-empty)
    best = [[INF] * (L + 1) for _ in range(num_stages + 1)]
    cut = [[-1] * (L + 1) for _ in range(num_stages + 1)]
    best[0][0] = 0.0
    for s in range(1, num_stages + 1):
        last = s == num_stages
        for i in range(s, L + 1):
            if last and i != L:
                continue
            for j in range(s - 1, i):
                if best[s - 1][j] == INF:
                    continue
                c = (i - j) + (head_cost if last else 0.0)
                v = max(best[s - 1][j], c)
                # strict '<' keeps the earliest cut => earlier stages take the extra layers
                if v < best[s][i] - 1e-12:
                    best[s][i] = v
                    cut[s][i] = j
    bounds, i = [], L
    for s in range(num_stages, 0, -1):
        j = cut[s][i]
        bounds.append((j, i))
        i = j
    bounds.reverse()
    return bounds


def balanced_split_blocks(spec: ModelSpec, num_stages: int, rows: Optional[int] = None) -> List[Tuple[int, int]]:
    """Like `balanced_split`, over attention / MLP blocks: returns half-open ranges of BLOCK indices
    (block 2 i = attention of layer i, 2 i + 1 = its MLP), minimising the heaviest stage. Exact DP."""
    U = 2 * spec.num_layers
    if num_stages == 1:
        return [(0, U)]
    if num_stages > U:
        raise ValueError(f"cannot split {U} blocks into {num_stages} stages")
    head_cost = head_cost_layers(spec)
    share = attn_block_share(spec, rows)
    cost = [share if u % 2 == 0 else 1.0 - share for u in range(U)]
    pre = [0.0]
    for c in cost:
        pre.append(pre[-1] + c)
    INF = float("inf")
    best = [[INF] * (U + 1) for _ in range(num_stages + 1)]
    cut = [[-1] * (U + 1) for _ in range(num_stages + 1)]
    best[0][0] = 0.0
    for s in range(1, num_stages + 1):
        last = s == num_stages
        for i in range(s, U + 1):
            if last and i != U:
                continue
            for j in range(s - 1, i):
                if best[s - 1][j] == INF:
                    continue
                v = max(best[s - 1][j], pre[i] - pre[j] + (head_cost if last else 0.0))
                if v < best[s][i] - 1e-12:
                    best[s][i] = v
                    cut[s][i] = j
    bounds, i = [], U
    for s in range(num_stages, 0, -1):
        j = cut[s][i]
        bounds.append((j, i))
        i = j
    bounds.reverse()
    return bounds


def plans_from_blocks(num_stages: int, bounds: Sequence[Tuple[int, int]]) -> List[StagePlan]:
    """Stage plans from half-open ranges of block indices (see `balanced_split_blocks`)."""
    out = []
    for s, (a, b) in enumerate(bounds):
        out.append(StagePlan(stage=s, num_stages=num_stages, layer_start=a // 2, layer_end=(b + 1) // 2,
                             has_embed=(s == 0), has_head=(s == num_stages - 1),
                             skip_attn_first=(a % 2 == 1), attn_only_last=(b % 2 == 1)))
    return out


def make_plans(spec: ModelSpec, num_stages: int, mode: str = "balanced",
               ranges: Optional[Sequence[Tuple[int, int]]] = None, split_layers: bool = False,
               rows: Optional[int] = None) -> List[StagePlan]:
    """Stage plans for `num_stages` stages.

    mode: "even" | "balanced" | "explicit" (with `ranges`).  Explicit ranges must tile
    [0, num_layers) contiguously with non-empty half-open intervals. `split_layers` (balanced mode):
    stage boundaries may fall between a layer's attention and MLP blocks; `rows`: sequences per
    micro-batch the pipeline will run (weighs those blocks: `attn_block_share`).
    """
    if mode == "balanced" and ranges is None and split_layers and num_stages > 1:
        return plans_from_blocks(num_stages, balanced_split_blocks(spec, num_stages, rows))
    if ranges is not None:
        mode = "explicit"
    if mode == "explicit":
        if ranges is None or len(ranges) != num_stages:
            raise ValueError("explicit partition needs one (start,end) range per stage")
        pos = 0
        for (a, b) in ranges:
            if a != pos or b <= a:
                raise ValueError(f"ranges must tile [0,{spec.num_layers}) contiguously, got {list(ranges)}")
            pos = b
        if pos != spec.num_layers:
            raise ValueError(f"ranges cover [0,{pos}) but the model has {spec.num_layers} layers")
        bounds = [tuple(r) for r in ranges]
    elif mode == "even":
        bounds = even_split(spec.num_layers, num_stages)
    elif mode == "balanced":
        bounds = balanced_split(spec, num_stages)
    else:
        raise ValueError(f"unknown partition mode {mode!r}")
    return [StagePlan(stage=s, num_stages=num_stages, layer_start=a, layer_end=b,
                      has_embed=(s == 0), has_head=(s == num_stages - 1))
            for s, (a, b) in enumerate(bounds)]


def make_worker_plans(spec: ModelSpec, num_workers: int, mode: str = "even",
                      ranges: Optional[Sequence[Tuple[int, int]]] = None):
    """The reference's topology: an orchestrator that owns embedding + final norm + lm_head and
    `num_workers` workers that own only decoder layers (orchestration.py:45-47; Worker1.py:68-70).
    Returns (edge_plan, worker_plans). With mode="even", 22 layers / 2 workers gives the
    reference's [0,11) and [11,22)."""
    if mode == "balanced":
        mode = "even"          # nothing but layers on the workers: balanced == even
    plans = make_plans(spec, num_workers, mode=mode, ranges=ranges)
    workers = [StagePlan(stage=p.stage, num_stages=num_workers, layer_start=p.layer_start,
                         layer_end=p.layer_end, has_embed=False, has_head=False) for p in plans]
    edge = StagePlan(stage=-1, num_stages=num_workers, layer_start=0, layer_end=0, has_embed=True, has_head=True)
    return edge, workers
