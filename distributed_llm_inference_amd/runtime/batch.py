"""Step descriptors: what the scheduler tells every pipeline stage to run.

A `StepBatch` is one micro-batch step over a ragged set of sequences: a prefill chunk has
`q_len > 1` new tokens for a sequence, a decode step has exactly one. The same descriptor
travels down the pipeline ahead of the activations (stage i forwards it to stage i+1), so
every stage scatters K/V into its own paged cache at the same (block, offset) coordinates and
no stage ever needs to ask another where a sequence lives.

The reference has no such thing: its `/process` message carries only a nested list of floats,
positions always restart at 0 and nothing identifies the sequence (Worker1.py:93-94,216-223),
which is why it must resend and recompute the whole sequence for every token.

Wire format (`to_bytes`/`from_bytes`): a fixed little-endian header followed by int32/float32
arrays — compact enough for a shared-memory ring slot, no pickling.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

KIND_STEP = 1        # run a forward step
KIND_RELEASE = 2     # sequences finished: forget their slots
KIND_SHUTDOWN = 3    # stop the stage loop
KIND_PING = 4        # liveness probe, forwarded down the pipe and acknowledged by the last stage
KIND_BARRIER = 5     # drain all outstanding work, then meet every other stage at a barrier

_MAGIC = 0x444C4942  # "DLIB"
_HEAD = "<IiiiiiiIii"   # magic, kind, step id, micro-batch, B, tokens (-1: none), table ints, flags, growth entries, live rows
_F_REPLAY, _F_COMPACT = 1, 2


@dataclass
class StepBatch:
    kind: int = KIND_STEP
    step_id: int = 0
    micro_batch: int = 0                 # which in-flight micro-batch (selects static buffers / graph)
    # per sequence (length B)
    slots: List[int] = field(default_factory=list)        # KV slot (row of the block table)
    q_lens: List[int] = field(default_factory=list)       # new tokens this step
    ctx_lens: List[int] = field(default_factory=list)     # tokens already in the cache
    sample: List[int] = field(default_factory=list)       # 1 -> produce a token for this sequence
    temperature: List[float] = field(default_factory=list)
    top_k: List[int] = field(default_factory=list)
    top_p: List[float] = field(default_factory=list)
    seeds: List[int] = field(default_factory=list)
    sample_steps: List[int] = field(default_factory=list)  # RNG counter (tokens generated so far)
    # block tables: only for sequences whose table changed since the stage last saw them
    table_slots: List[int] = field(default_factory=list)
    table_blocks: List[List[int]] = field(default_factory=list)
    # on-demand growth: block_table[grow_slots[i]][grow_pos[i]] = grow_blocks[i] (a sequence that
    # crossed a 32-token boundary got one more block); applied before the step runs, replay or not
    grow_slots: List[int] = field(default_factory=list)
    grow_pos: List[int] = field(default_factory=list)
    grow_blocks: List[int] = field(default_factory=list)
    # stage-0 input: explicit token ids (prefill, or decode whose token the scheduler knows);
    # None => take the tokens the last stage fed back for this micro-batch
    token_ids: Optional[List[int]] = None
    # decode fast path: all q_lens == 1 and the composition is unchanged since the previous
    # step of this micro-batch, so stages may replay their captured graph as is
    replay: bool = False
    # Compact replay descriptor: the per-sequence arrays are left out altogether (a stage replaying
    # its captured graph reads none of them; at 2048 rows they are 80 KB and a millisecond of
    # packing per stage and step). `rows` = sequences of the micro-batch, `live` = those that decode.
    rows: int = -1
    live: int = -1

    @property
    def compact(self) -> bool:
        return self.rows >= 0

    @property
    def num_seqs(self) -> int:
        return self.rows if self.rows >= 0 else len(self.slots)

    @property
    def num_tokens(self) -> int:
        return self.live if self.rows >= 0 else int(sum(self.q_lens))

    @property
    def is_decode(self) -> bool:
        """Single-token step (rows with q_len 0 are padding of a fixed-shape micro-batch)."""
        if self.rows >= 0:
            return self.live > 0
        return bool(self.q_lens) and all(q in (0, 1) for q in self.q_lens) and any(q == 1 for q in self.q_lens)

    @property
    def num_samples(self) -> int:
        return self.live if self.rows >= 0 else int(sum(self.sample))

    @staticmethod
    def replay_of(step_id: int, micro_batch: int, rows: int, live: int) -> "StepBatch":
        """The descriptor of "the same micro-batch once more" (only valid for stages that run
        fixed-shape decode steps over static buffers; growth entries may be added)."""
        return StepBatch(step_id=step_id, micro_batch=micro_batch, replay=True, rows=rows, live=live)

    def validate(self):
        if self.compact:
            if not self.replay or self.token_ids is not None or self.table_slots or self.slots:
                raise ValueError("a compact descriptor is a bare replay")
            if not (len(self.grow_slots) == len(self.grow_pos) == len(self.grow_blocks)):
                raise ValueError("grow_slots / grow_pos / grow_blocks mismatch")
            return
        B = self.num_seqs
        for name in ("q_lens", "ctx_lens", "sample", "temperature", "top_k", "top_p", "seeds", "sample_steps"):
            if len(getattr(self, name)) != B:
                raise ValueError(f"StepBatch.{name} has {len(getattr(self, name))} entries for {B} sequences")
        if self.token_ids is not None and len(self.token_ids) != self.num_tokens and not (
                self.is_decode and len(self.token_ids) == B):     # fixed-shape decode: one id per row
            raise ValueError("token_ids length must equal the number of new tokens")
        if len(self.table_slots) != len(self.table_blocks):
            raise ValueError("table_slots / table_blocks mismatch")
        if not (len(self.grow_slots) == len(self.grow_pos) == len(self.grow_blocks)):
            raise ValueError("grow_slots / grow_pos / grow_blocks mismatch")

    # ---- wire format ------------------------------------------------------------------
    def to_bytes(self) -> bytes:
        if self.compact:
            ng = len(self.grow_slots)
            head = struct.pack(_HEAD, _MAGIC, self.kind, self.step_id, self.micro_batch, self.rows, -1, 0,
                               _F_REPLAY | _F_COMPACT, ng, self.live)
            return head + np.asarray(self.grow_slots + self.grow_pos + self.grow_blocks, dtype="<i4").tobytes()
        B = self.num_seqs
        ntok = -1 if self.token_ids is None else len(self.token_ids)
        tb_flat: List[int] = []
        for s, blocks in zip(self.table_slots, self.table_blocks):
            tb_flat += [s, len(blocks)] + list(blocks)
        ng = len(self.grow_slots)
        head = struct.pack(_HEAD, _MAGIC, self.kind, self.step_id, self.micro_batch, B, ntok,
                           len(tb_flat), _F_REPLAY if self.replay else 0, ng, -1)
        ints = np.asarray(self.slots + self.q_lens + self.ctx_lens + self.sample + self.top_k +
                          self.sample_steps + tb_flat + self.grow_slots + self.grow_pos + self.grow_blocks +
                          (self.token_ids or []), dtype="<i4")
        flts = np.asarray(self.temperature + self.top_p, dtype="<f4")
        seeds = np.asarray(self.seeds, dtype="<i8")
        return head + ints.tobytes() + flts.tobytes() + seeds.tobytes()

    @staticmethod
    def from_bytes(buf: bytes) -> "StepBatch":
        hs = struct.calcsize(_HEAD)
        magic, kind, step_id, mb, B, ntok, ntb, flags, ng, live = struct.unpack_from(_HEAD, buf, 0)
        if magic != _MAGIC:
            raise ValueError("bad step descriptor magic")
        replay = flags & _F_REPLAY
        if flags & _F_COMPACT:
            g = np.frombuffer(buf, dtype="<i4", count=3 * ng, offset=hs).tolist()
            return StepBatch(kind=kind, step_id=step_id, micro_batch=mb, replay=True, rows=B, live=live,
                             grow_slots=g[:ng], grow_pos=g[ng:2 * ng], grow_blocks=g[2 * ng:])
        n_int = 6 * B + ntb + 3 * ng + max(ntok, 0)
        ints = np.frombuffer(buf, dtype="<i4", count=n_int, offset=hs).tolist()
        off = hs + 4 * n_int
        flts = np.frombuffer(buf, dtype="<f4", count=2 * B, offset=off).tolist()
        off += 8 * B
        seeds = np.frombuffer(buf, dtype="<i8", count=B, offset=off).tolist()
        f = [ints[i * B:(i + 1) * B] for i in range(6)]
        tb = ints[6 * B:6 * B + ntb]
        table_slots, table_blocks, i = [], [], 0
        while i < len(tb):
            s, n = tb[i], tb[i + 1]
            table_slots.append(s)
            table_blocks.append(tb[i + 2:i + 2 + n])
            i += 2 + n
        g0 = 6 * B + ntb
        grow = [ints[g0 + i * ng:g0 + (i + 1) * ng] for i in range(3)]
        toks = ints[g0 + 3 * ng:] if ntok >= 0 else None
        return StepBatch(kind=kind, step_id=step_id, micro_batch=mb, slots=f[0], q_lens=f[1], ctx_lens=f[2],
                         sample=f[3], top_k=f[4], sample_steps=f[5], temperature=flts[:B], top_p=flts[B:],
                         seeds=seeds, table_slots=table_slots, table_blocks=table_blocks, token_ids=toks,
                         replay=bool(replay), grow_slots=grow[0], grow_pos=grow[1], grow_blocks=grow[2])


def control(kind: int, slots: Optional[List[int]] = None, step_id: int = 0) -> StepBatch:
    n = len(slots or [])
    return StepBatch(kind=kind, step_id=step_id, slots=list(slots or []), q_lens=[0] * n, ctx_lens=[0] * n,
                     sample=[0] * n, temperature=[0.0] * n, top_k=[0] * n, top_p=[1.0] * n, seeds=[0] * n,
                     sample_steps=[0] * n)
