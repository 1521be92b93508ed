"""Request scheduler and generation driver (runs on pipeline rank 0).

One thread owns all mutable serving state: the request queue, the KV block tables and the
micro-batches. HTTP handlers only enqueue requests and wait on futures. (The reference runs
`generate_with_sampling` directly inside Flask's threaded dev server on shared globals with no
lock: orchestration.py:234,333,372.)

Scheduling model — continuous batching over `num_micro_batches` fixed-shape micro-batches:

* a request is admitted into a free row of a micro-batch when the KV pool has blocks for its
  prompt (plus a small watermark), prefilled in chunks of at most `prefill_chunk` tokens, then
  joins the micro-batch's decode steps; its block table grows one 32-token block at a time as it
  generates, and when the pool runs dry the youngest sequences of the starved micro-batch are
  preempted — blocks released, request requeued at the front, prompt + tokens so far prefilled
  again on re-admission — so the pool is sized from HBM, not from slots x max_seq_len;
* steady-state decode steps are "replays": the token fed to stage 0 is the one the last stage
  sampled for that row (device-to-device feedback), context lengths and RNG counters advance
  on the device, and up to `max_inflight_steps` steps of a micro-batch are issued before their
  results are read — so with >= pp_size micro-batches in flight every stage always has work,
  which is what removes the pipeline bubble the reference has by construction (one request,
  strictly sequential hops: orchestration.py:109-137);
* a micro-batch is recomposed (drained, finished rows retired, waiting requests admitted and
  prefilled) only when that is worth a bubble: waiting work and free/dead rows, or all rows dead.

The autoregressive semantics are the reference's (orchestration.py:69-228): stop at EOS without
emitting it, cap at max_new_tokens, temperature/top-k/top-p sampling; and so is the result
dictionary of `generate_text` (orchestration.py:211-218) and its error envelope (:220-228).
"""
from __future__ import annotations

import collections
import threading
import time
import traceback
from dataclasses import dataclass, field
from typing import Deque, Dict, List, Optional


from ..config import EngineConfig
from .batch import KIND_RELEASE, StepBatch, control
from .head import LocalHead
from .kv_cache import BlockAllocator, blocks_for
from ..ops.packing import KV_BLOCK
from .sampler import SamplingParams
from .tokenizer import build_prompt, encode_prompt


class EngineError(RuntimeError):
    pass


class Histogram:
    """Cumulative histogram with fixed upper bounds (seconds), in the shape Prometheus expects."""

    def __init__(self, bounds):
        self.bounds = list(bounds)
        self.counts = [0] * (len(self.bounds) + 1)      # last: +Inf
        self.total, self.n = 0.0, 0

    def observe(self, v: float):
        import bisect
        self.counts[bisect.bisect_left(self.bounds, v)] += 1
        self.total += v
        self.n += 1

    def snapshot(self):
        cum, out = 0, []
        for b, c in zip(self.bounds + [float("inf")], self.counts):
            cum += c
            out.append((b, cum))
        return {"buckets": out, "sum": self.total, "count": self.n}


TTFT_BOUNDS = (0.005, 0.01, 0.025, 0.05, 0.1, 0.25, 0.5, 1.0, 2.5, 5.0, 10.0, 30.0)
ITL_BOUNDS = (0.001, 0.002, 0.004, 0.008, 0.016, 0.032, 0.064, 0.128, 0.256, 0.512, 1.0, 4.0)


@dataclass
class Request:
    rid: int
    prompt_ids: List[int]
    params: SamplingParams
    eos_id: Optional[int]
    stop_ids: frozenset = frozenset()     # every id that ends the sequence (eos_id and the spec's further ones)
    t_arrival: float = field(default_factory=time.perf_counter)
    state: str = "waiting"            # waiting | prefill | running | finished | failed
    mb: int = -1
    row: int = -1
    slot: int = -1
    blocks: List[int] = field(default_factory=list)
    context: List[int] = field(default_factory=list)   # tokens to prefill: prompt (+ output so far after a preemption)
    preemptions: int = 0
    prefilled: int = 0                # context tokens submitted so far
    issued_ctx: int = 0               # tokens whose KV the submitted steps (will) have cached
    issued_samples: int = 0           # sampling steps submitted
    output: List[int] = field(default_factory=list)
    token_times: List[float] = field(default_factory=list)
    t_first: float = 0.0
    t_done: float = 0.0
    error: Optional[str] = None
    done: threading.Event = field(default_factory=threading.Event)
    seed: int = 0
    moved: bool = False               # its row changed since its block table was last sent to the stages
    callbacks: List = field(default_factory=list)     # called (from the scheduler thread) when the request is done
    token_callbacks: List = field(default_factory=list)   # called (from the scheduler thread) after each output token

    def _set_done(self):
        self.done.set()
        for cb in self.callbacks:
            try:
                cb(self)
            except Exception:
                pass

    def add_done_callback(self, cb):
        """`cb(request)` runs once the request has finished or failed (immediately if it already has)."""
        self.callbacks.append(cb)
        if self.done.is_set():
            cb(self)

    @property
    def finished(self) -> bool:
        return self.state in ("finished", "failed")

    def wait(self, timeout: Optional[float] = None) -> "Request":
        if not self.done.wait(timeout):
            raise TimeoutError(f"request {self.rid} timed out after {timeout}s")
        return self


class _Inflight:
    """One submitted step: its descriptor, the requests of its rows, and the tokens on their way."""

    def __init__(self, desc: StepBatch, rows: List[Optional[Request]], res):
        self.desc, self.rows, self.res = desc, rows, res
        self.t_submit = time.perf_counter()

    def ready(self) -> bool:
        return self.res.ready()

    def get(self) -> List[int]:
        return self.res.get()


class _MicroBatch:
    def __init__(self, idx: int, size: int):
        self.idx = idx
        self.rows: List[Optional[Request]] = [None] * size
        self.inflight: Deque[_Inflight] = collections.deque()
        self.state = "idle"           # idle | steady | draining | prefill
        self.fresh: List[Request] = []    # admitted, prefill not finished
        self.nrows = size                 # rows a decode step of the current composition covers (a graph size)
        self.starved = False          # a decode step could not be issued: the KV pool had no block for a row

    def live(self) -> List[Request]:
        return [r for r in self.rows if r is not None and r.state == "running"]

    def dead_rows(self) -> int:
        return sum(1 for r in self.rows if r is not None and r.finished)

    def free_rows(self) -> int:
        return sum(1 for r in self.rows if r is None)


class Engine:
    def __init__(self, cfg: EngineConfig, head, tokenizer, num_blocks: int, board=None, tracer=None):
        """`head`: the scheduler's handle on pipeline stage 0 — a `LocalHead` / `RemoteHead`, or a
        `StageRuntime` of this process (wrapped in a `LocalHead`). `num_blocks`: KV blocks every
        stage's cache holds (the minimum over the stages' HBM budgets)."""
        if not hasattr(head, "submit"):
            head = LocalHead(head)
        self.cfg, self.head, self.tok = cfg, head, tokenizer
        self.rt = getattr(head, "rt", None)
        self.spec = head.spec
        self.num_stages = head.num_stages
        self.board, self.tracer = board, tracer
        self.alloc = BlockAllocator(num_blocks)
        self.max_blocks_per_seq = blocks_for(cfg.max_seq_len)
        # admission keeps this many blocks free for the growth of the sequences already running
        self.watermark = min(max(1, num_blocks // 100), max(0, num_blocks - self.max_blocks_per_seq))
        self.mbs = [_MicroBatch(i, cfg.max_batch) for i in range(cfg.num_micro_batches)]
        self.waiting: Deque[Request] = collections.deque()
        self._inbox: Deque[Request] = collections.deque()
        self._cancel_box: Deque[Request] = collections.deque()
        self._lock = threading.Lock()
        self._wake = threading.Event()
        self._stop = False
        self._thread: Optional[threading.Thread] = None
        self._rid = 0
        self._step_id = 0
        self.fatal: Optional[str] = None
        self.on_gpu = head.on_gpu
        self._rr = 0
        # Fixed-shape decode (static buffers, hipGraph) keeps computing a finished row as padding and
        # feeds every row's token back on the device; an eager executor feeds back one token per
        # LIVE row, so there a replay step is only valid while the set of live rows is unchanged.
        B = cfg.max_batch
        probe = StepBatch(slots=list(range(B)), q_lens=[1] * B, ctx_lens=[0] * B, sample=[1] * B, temperature=[0.0] * B,
                          top_k=[1] * B, top_p=[1.0] * B, seeds=[0] * B, sample_steps=[0] * B)
        self._static_decode = bool(head.graph_decode(probe))
        self._graph_sizes = cfg.graph_sizes() if self._static_decode else [B]
        # counters for /metrics
        self.stats = {"requests": 0, "completed": 0, "failed": 0, "tokens_out": 0, "tokens_in": 0,
                      "steps": 0, "prefill_steps": 0, "decode_steps": 0, "recompositions": 0,
                      "preemptions": 0, "kv_blocks_grown": 0, "cancelled": 0, "rejected": 0, "rows_moved": 0,
                      "rider_tokens": 0}
        self.latencies: Deque[float] = collections.deque(maxlen=4096)     # per-token decode latencies (s)
        self.ttfts: Deque[float] = collections.deque(maxlen=1024)
        self.hist_ttft = Histogram(TTFT_BOUNDS)          # arrival -> first token
        self.hist_itl = Histogram(ITL_BOUNDS)            # gap between consecutive tokens of a request

    # ---- lifecycle -------------------------------------------------------------------------
    def start(self):
        if self._thread is None:
            self._thread = threading.Thread(target=self._loop, name="dlia-scheduler", daemon=True)
            self._thread.start()
        return self

    def stop(self, timeout: float = 10.0):
        self._stop = True
        self._wake.set()
        if self._thread is not None:
            self._thread.join(timeout)
            self._thread = None
        if self.fatal is None:
            self.head.shutdown()          # after a fatal error the stages are replaced, not drained

    # ---- public API ------------------------------------------------------------------------
    def submit(self, prompt_ids: List[int], params: Optional[SamplingParams] = None) -> Request:
        params = params or SamplingParams(self.cfg.temperature, self.cfg.top_k, self.cfg.top_p,
                                          self.cfg.default_max_tokens)
        with self._lock:
            self._rid += 1
            rid = self._rid
        eos = getattr(self.tok, "eos_token_id", None)
        if eos is None:
            eos = self.spec.eos_token_id
        stop = frozenset(i for i in (eos, *getattr(self.spec, "extra_eos_token_ids", ())) if i is not None)
        req = Request(rid=rid, prompt_ids=list(prompt_ids), params=params, eos_id=eos, stop_ids=stop)
        req.context = req.prompt_ids
        req.seed = params.seed if params.seed is not None else (time.time_ns() ^ (rid * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF
        if self.fatal:
            self._fail(req, self.fatal)
            return req
        if len(req.prompt_ids) == 0:
            self._fail(req, "empty prompt")
            return req
        if len(req.prompt_ids) + 1 > self.cfg.max_seq_len:
            self._fail(req, f"prompt of {len(req.prompt_ids)} tokens exceeds max_seq_len {self.cfg.max_seq_len}")
            return req
        if self.cfg.max_queue > 0 and len(self.waiting) + len(self._inbox) >= self.cfg.max_queue:
            # back-pressure: answering "try again" now beats an answer after the client's own deadline
            self.stats["rejected"] += 1
            self._fail(req, f"queue full ({self.cfg.max_queue} requests waiting)")
            return req
        self.stats["requests"] += 1
        self.stats["tokens_in"] += len(req.prompt_ids)
        self._inbox.append(req)
        self._wake.set()
        return req

    def cancel(self, req: Request):
        """Abandon `req` (its client went away or stopped waiting): from any thread. The scheduler
        marks it failed with error "cancelled" at its next tick; a queued request leaves the queue,
        a running one stops receiving tokens at once and gives its row and KV blocks back at the
        micro-batch's next recomposition (until then the row rides along as padding, like a row
        that hit EOS). The reference has no way to stop a generation (orchestration.py:131-199
        runs its loop to the end whoever is still listening)."""
        if req.finished:
            return
        self._cancel_box.append(req)
        self._wake.set()

    def submit_text(self, prompt: str, max_new_tokens: Optional[int] = None, temperature: Optional[float] = None,
                    top_k: Optional[int] = None, top_p: Optional[float] = None, seed: Optional[int] = None) -> Request:
        """Chat-format and tokenise `prompt`, clamp the token count, enqueue. The caller waits on the
        returned request (`wait`, or `add_done_callback` from an event loop) and renders it with
        `text_result`."""
        text = build_prompt(self.tok, prompt, self.cfg.system_prompt, self.cfg.chat_template)
        ids = encode_prompt(self.tok, text)
        n = self.cfg.default_max_tokens if max_new_tokens is None else int(max_new_tokens)
        n = max(1, min(n, self.cfg.max_tokens_cap))
        sp = SamplingParams(temperature=self.cfg.temperature if temperature is None else float(temperature),
                            top_k=self.cfg.top_k if top_k is None else int(top_k),
                            top_p=self.cfg.top_p if top_p is None else float(top_p),
                            max_new_tokens=n, seed=seed)
        return self.submit(ids, sp)

    def text_result(self, req: Request, prompt: str, start: float) -> Dict:
        """The reference's result dictionary (orchestration.py:211-218) for a finished request, or
        its error envelope (:220-228) for a failed one."""
        if req.state == "failed":
            return {"error": f"Error: {req.error or 'generation failed'}", "status": "failed"}
        generated_text = self.tok.decode(req.output, skip_special_tokens=True)
        elapsed = time.time() - start
        tps = len(req.output) / elapsed if elapsed > 0 else 0
        out = {"prompt": prompt, "response": generated_text, "status": "success",
               "time_taken": f"{elapsed:.2f}s", "tokens_generated": len(req.output),
               "tokens_per_sec": f"{tps:.2f}"}
        # numeric extras (the reference's three metrics above are strings)
        out["ttft_ms"] = round((req.t_first - req.t_arrival) * 1e3, 3) if req.t_first else None
        if len(req.token_times) > 1:
            gaps = sorted(b - a for a, b in zip(req.token_times, req.token_times[1:]))
            out["tpot_ms_p50"] = round(gaps[len(gaps) // 2] * 1e3, 3)
        if self.cfg.debug_top5:
            out["token_ids"] = list(req.output)
        return out

    def generate_text(self, prompt: str, max_new_tokens: Optional[int] = None, temperature: Optional[float] = None,
                      top_k: Optional[int] = None, top_p: Optional[float] = None, seed: Optional[int] = None,
                      timeout: Optional[float] = None) -> Dict:
        """Prompt in, text out, with the reference's result / error dictionaries
        (orchestration.py:211-228)."""
        start = time.time()
        req = None
        try:
            req = self.submit_text(prompt, max_new_tokens, temperature, top_k, top_p, seed)
            req.wait(timeout or self.cfg.request_timeout_s)
            return self.text_result(req, prompt, start)
        except Exception as e:      # same envelope as the reference: HTTP 200, status "failed"
            if req is not None:
                self.cancel(req)    # nobody waits for it any more
            return {"error": f"Error: {e}", "status": "failed"}

    # ---- scheduler thread ------------------------------------------------------------------
    def _loop(self):
        try:
            # HIP's current device is per thread: this thread drives stage 0's streams, graphs and kernels
            rt = getattr(self.head, "rt", None)
            if rt is not None and getattr(rt, "on_gpu", False):
                import torch
                dev = torch.device(rt.device)
                if dev.index is not None:
                    torch.cuda.set_device(dev)
            while not self._stop:
                progressed = False
                while self._inbox:
                    req = self._inbox.popleft()
                    if not req.finished:            # cancelled between submit() and this tick: never queued
                        self.waiting.append(req)
                    progressed = True
                while self._cancel_box:
                    self._cancel(self._cancel_box.popleft())
                    progressed = True
                n = len(self.mbs)
                for i in range(n):
                    mb = self.mbs[(self._rr + i) % n]
                    progressed |= self._tick(mb)
                self._rr = (self._rr + 1) % n
                self._check_pipeline()
                if not progressed:
                    busy = any(mb.inflight or mb.state != "idle" for mb in self.mbs) or self.waiting
                    if busy:
                        time.sleep(20e-6)
                    else:
                        self._wake.wait(0.05)
                        self._wake.clear()
        except BaseException as e:
            traceback.print_exc()
            self._fatal(f"{type(e).__name__}: {e}")

    def _check_pipeline(self):
        """Watchdog: a stage that died or reported an error fails everything in flight."""
        if self.board is None or (self.num_stages == 1 and not self.head.remote):
            return
        now = time.perf_counter()
        if now - getattr(self, "_last_check", 0.0) < 0.25:
            return
        self._last_check = now
        from ..utils.procs import pid_alive
        for s in range(0 if self.head.remote else 1, self.num_stages):
            st = self.board.read(s)
            dead = False
            if st["state"] == "error":
                dead = True
                why = f"stage {s} failed: {st['error']}"
            elif st["state"] in ("ready", "starting") and st["pid"] > 0 and not pid_alive(st["pid"]):
                dead, why = True, f"stage {s} (pid {st['pid']}) is gone"
            if not dead and any(mb.inflight for mb in self.mbs):
                oldest = min(r.t_submit for mb in self.mbs for r in mb.inflight)
                if now - oldest > self.cfg.hop_timeout_s * self.num_stages:
                    dead, why = True, f"pipeline stalled: no result for {now - oldest:.1f}s"
            if dead:
                raise EngineError(why)

    def _fatal(self, msg: str):
        self.fatal = msg
        for mb in self.mbs:
            for r in mb.rows:
                if r is not None and not r.finished:
                    self._fail(r, msg)
            mb.inflight.clear()
        for r in list(self.waiting) + list(self._inbox):
            self._fail(r, msg)
        self.waiting.clear()

    def _cancel(self, req: Request):
        if req.finished:
            return
        if req.state == "waiting":
            try:                           # (a request still in the inbox is dropped when the inbox is drained)
                self.waiting.remove(req)
            except ValueError:
                pass
        # a request that holds a row becomes a dead row: skipped by `_harvest`, not grown, retired
        # (blocks released, slot freed on the stages) when its micro-batch is next recomposed
        req.error, req.state = "cancelled", "failed"
        req.t_done = time.perf_counter()
        self.stats["cancelled"] += 1
        if self.tracer is not None:
            self.tracer.instant("scheduler", "cancel", {"rid": req.rid, "tokens": len(req.output)})
        req._set_done()

    def _fail(self, req: Request, msg: str):
        req.error, req.state = msg, "failed"
        req.t_done = time.perf_counter()
        self.stats["failed"] += 1
        req._set_done()

    # ---- per-micro-batch state machine -----------------------------------------------------
    def _tick(self, mb: _MicroBatch) -> bool:
        progressed = False
        while mb.inflight and mb.inflight[0].ready():
            self._harvest(mb, mb.inflight.popleft())
            progressed = True

        if mb.state == "idle":
            if self.waiting:
                mb.state = "draining"
            else:
                return progressed

        if mb.state == "steady":
            live = mb.live()
            waiting_fits = self._worth_stopping_for(mb)
            if (not live or waiting_fits or mb.dead_rows() * 2 >= max(1, len(live) + mb.dead_rows())
                    or (mb.dead_rows() > 0 and not self._static_decode)):
                mb.state = "draining"
            elif (len(mb.inflight) < self.cfg.max_inflight_steps and self._budget_left(mb)
                  and self.head.fb_ready(mb.idx)):
                if self._submit_decode(mb, replay=True):
                    return True
                mb.state, mb.starved = "draining", True       # no KV block for a row: recompose

        if mb.state == "draining":
            if mb.inflight:
                return progressed
            self._retire(mb)
            self._compact(mb)
            if mb.starved:
                if not self._make_room(mb):
                    return progressed          # other micro-batches hold the pool: wait for them
                mb.starved = False
            else:
                self._admit(mb)
            if mb.fresh:
                mb.state = "prefill"
            elif mb.live():
                if self._submit_decode(mb, replay=False):
                    mb.state = "steady"
                else:
                    mb.starved = True
                return True
            else:
                mb.state = "idle"
                return True

        if mb.state == "prefill":
            if mb.inflight:
                return progressed
            pending = [r for r in mb.fresh if r.prefilled < len(r.context) and not r.finished]
            if pending:
                self._submit_prefill(mb, pending)
                return True
            mb.fresh = []
            if mb.live():
                if self._submit_decode(mb, replay=False):
                    mb.state = "steady"
                else:
                    mb.state, mb.starved = "draining", True
            else:
                mb.state = "draining"
            return True
        return progressed

    def _budget_left(self, mb: _MicroBatch) -> bool:
        """Is there a row that still needs tokens beyond what is already in flight?"""
        for r in mb.live():
            if self._wants_more(r):
                return True
        return False

    def _wants_more(self, r: Request) -> bool:
        return r.issued_samples < r.params.max_new_tokens and r.issued_ctx + 1 <= self.cfg.max_seq_len

    def _retire(self, mb: _MicroBatch):
        gone = []
        for i, r in enumerate(mb.rows):
            if r is not None and r.finished:
                self.alloc.release(r.rid)
                gone.append(r.slot)
                mb.rows[i] = None
        if gone:
            self.stats["recompositions"] += 1
            self._dispatch(control(KIND_RELEASE, gone))

    MOVE_BUDGET = 2048      # block-table entries one recomposition may re-send for moved rows (they ride on one descriptor)

    def _compact(self, mb: _MicroBatch):
        """With several decode step sizes: move the running rows of a drained micro-batch down into
        the holes below them when that lets its next steps run at a smaller size — a straggler in row
        2000 of an otherwise empty micro-batch would keep every step at 2048 rows. A row is a slot of
        the stages' block table, nothing else: the old slots are released and the next (non-replay)
        decode step carries the moved requests' tables under their new slots; no KV data moves."""
        sizes = self._graph_sizes
        if len(sizes) < 2:
            return
        live = [r for r in mb.rows if r is not None]
        n = len(live)
        if n == 0 or n == len(mb.rows):
            return
        size_of = lambda rows: next((b for b in sizes if b >= rows), len(mb.rows))
        if size_of(n) >= size_of(max(r.row for r in live) + 1):
            return                                   # packing them would not reach a smaller step
        movers = sorted((r for r in live if r.row >= n), key=lambda r: r.row)
        if sum(len(r.blocks) for r in movers) > self.MOVE_BUDGET:
            return
        holes = [i for i in range(n) if mb.rows[i] is None]
        old = []
        for r, i in zip(movers, holes):
            old.append(r.slot)
            mb.rows[r.row] = None
            r.row, r.slot, r.moved = i, mb.idx * self.cfg.max_batch + i, True
            mb.rows[i] = r
        self.stats["rows_moved"] += len(old)
        self._dispatch(control(KIND_RELEASE, old))

    def _worth_stopping_for(self, mb: _MicroBatch) -> bool:
        """Should this running micro-batch be recomposed to take waiting requests in? Yes when they
        would fill a worthwhile share of its rows, or when the oldest has waited long enough."""
        if not self.waiting:
            return False
        room = mb.free_rows() + mb.dead_rows()
        if room <= 0 or not self._could_admit():
            return False
        fill = min(len(self.waiting), room)
        if fill >= max(1, int(self.cfg.admit_fill_fraction * self.cfg.max_batch)):
            return True
        return (time.perf_counter() - self.waiting[0].t_arrival) * 1e3 >= self.cfg.admit_max_wait_ms

    # ---- KV blocks: admission, growth, preemption ----------------------------------------------
    def _could_admit(self) -> bool:
        """Would the request at the head of the queue get its blocks right now?"""
        if not self.waiting:
            return False
        need = max(1, blocks_for(len(self.waiting[0].context)))
        return self.alloc.free - need >= self.watermark or self.alloc.used == 0

    def _admit(self, mb: _MicroBatch):
        while self.waiting and mb.free_rows() > 0:
            req = self.waiting[0]
            if req.finished:               # cancelled while queued: takes no row and no blocks
                self.waiting.popleft()
                continue
            need = max(1, blocks_for(len(req.context)))
            if need > self.max_blocks_per_seq or need > self.alloc.num_blocks:
                self.waiting.popleft()
                self._fail(req, "request needs more KV blocks than the cache holds")
                continue
            if self.alloc.used > 0 and self.alloc.free - need < self.watermark:
                break                      # keep room for the running sequences to grow
            blocks = self.alloc.take(req.rid, need)
            if blocks is None:
                break                      # wait for running requests to free blocks
            self.waiting.popleft()
            row = mb.rows.index(None)
            req.mb, req.row, req.slot = mb.idx, row, mb.idx * self.cfg.max_batch + row
            req.blocks, req.state = blocks, "prefill"
            req.prefilled = req.issued_ctx = 0
            mb.rows[row] = req
            mb.fresh.append(req)
            if self.tracer is not None:
                self.tracer.instant("scheduler", "admit", {"rid": req.rid, "mb": mb.idx, "row": row,
                                                           "prompt_tokens": len(req.context)})

    def _needs_block(self, r: Request) -> bool:
        """Will the next decode step of `r` write its K/V beyond the blocks it owns?"""
        return self._wants_more(r) and r.issued_ctx // KV_BLOCK >= len(r.blocks)

    def _grow(self, mb: _MicroBatch, d: Optional[StepBatch]) -> bool:
        """One more block for every live row about to cross a block boundary; the new table entries
        ride on `d`. All or nothing: False (and nothing taken) when the pool cannot serve all."""
        need = [r for r in mb.live() if self._needs_block(r)]
        if not need:
            return True
        if self.alloc.free < len(need):
            return False
        for r in need:
            blk = self.alloc.take(r.rid, 1)
            if blk is None:                # cannot happen after the check above; stay consistent anyway
                return False
            if d is not None:
                d.grow_slots.append(r.slot); d.grow_pos.append(len(r.blocks)); d.grow_blocks.append(blk[0])
            r.blocks.append(blk[0])
            self.stats["kv_blocks_grown"] += 1
        return True

    def _preempt(self, mb: _MicroBatch, r: Request):
        """Give `r`'s blocks back and requeue it at the front; on re-admission its prompt and the
        tokens generated so far are prefilled again (recompute instead of swapping to the host)."""
        self.alloc.release(r.rid)
        self._dispatch(control(KIND_RELEASE, [r.slot]))
        mb.rows[r.row] = None
        if r in mb.fresh:
            mb.fresh.remove(r)
        r.context = r.prompt_ids + r.output
        r.blocks, r.state, r.mb, r.row, r.slot, r.moved = [], "waiting", -1, -1, -1, False
        r.prefilled = r.issued_ctx = 0
        r.issued_samples = len(r.output)
        r.preemptions += 1
        self.stats["preemptions"] += 1
        self.waiting.appendleft(r)
        if self.tracer is not None:
            self.tracer.instant("scheduler", "preempt", {"rid": r.rid, "tokens": len(r.output)})

    def _make_room(self, mb: _MicroBatch) -> bool:
        """A drained micro-batch whose rows could not all grow: preempt its youngest sequences until
        the rest can. The oldest running request of the whole engine is never the victim, so some
        request always makes progress. False: nothing can be done from this micro-batch right now."""
        while True:
            need = sum(1 for r in mb.live() if self._needs_block(r))
            if self.alloc.free >= need:
                return True
            live = sorted(mb.live(), key=lambda r: r.rid)
            oldest_anywhere = min((r.rid for m in self.mbs for r in m.live()), default=None)
            victims = [r for r in live if r.rid != oldest_anywhere]
            if not victims:
                only = live[0] if live else None
                if only is not None and self.alloc.used == len(only.blocks):
                    self._fail(only, "request needs more KV blocks than the cache holds")
                    self._retire(mb)
                    continue
                return False
            self._preempt(mb, victims[-1])

    # ---- step construction -------------------------------------------------------------------
    def _next_step_id(self) -> int:
        self._step_id += 1
        return self._step_id

    NEWCOMERS_RIDE_UP_TO = 512      # rows per micro-batch up to which the newcomers of one admission ride in its later chunks

    def _submit_prefill(self, mb: _MicroBatch, pending: List[Request]):
        """One ragged step: prompt chunks of the micro-batch's newcomers and — `mixed_prefill` — one
        decode token for each of its running sequences, so that a recomposition does not leave those
        standing still for as long as the newcomers' prompts take (the reference has neither: one
        request at a time, whole sequence per step, orchestration.py:109-111)."""
        budget = self.cfg.prefill_chunk
        d = StepBatch(step_id=self._next_step_id(), micro_batch=mb.idx, token_ids=[])
        rows: List[Optional[Request]] = []
        # Riders: the sequences that were decoding before this admission, and — on micro-batches of up to
        # NEWCOMERS_RIDE_UP_TO rows — also the admission's own newcomers once their prompt is in. A token that rides in
        # a 4096-token ragged step costs what a prompt token costs (~11 us on Llama-3-8B); a token of a captured
        # decode step costs the step's time over its rows: 47 us at 256 rows (weight streaming), 15 us at 2048
        # (compute-bound). So on small micro-batches every ridden token is a cheap one (closed loop of 512 clients
        # on 2 x 256 rows: 335 decode steps instead of 565, inter-token p99 70 instead of 163 ms), while a 2048-row micro-batch refilled
        # in one go (60 chunks) would only drag a growing tail of single tokens through eager steps and lose its
        # common start (8192 requests at once on 2 x 2048 rows: 33.9 k against 38.4 k output tok/s).
        newcomers = {id(r) for r in mb.fresh} if self.cfg.max_batch > self.NEWCOMERS_RIDE_UP_TO else ()
        riders = [r for r in mb.rows if r is not None and r.state == "running" and self._wants_more(r) and r.output
                  and id(r) not in newcomers] if self.cfg.mixed_prefill else []
        # (nothing of this micro-batch is in flight here, so every rider's last token is known; riders need
        # room in the step next to a worthwhile chunk, a KV block where they cross a boundary, and at least one
        # real prompt chunk beside them — a step of single tokens only would be a decode step)
        if riders and (len(riders) > budget // 2 or all(len(r.context) - r.prefilled == 1 for r in pending)
                       or not self._grow(mb, d)):
            riders = []
        for r in riders:
            d.slots.append(r.slot); d.q_lens.append(1); d.ctx_lens.append(r.issued_ctx); d.sample.append(1)
            d.temperature.append(r.params.temperature); d.top_k.append(r.params.top_k); d.top_p.append(r.params.top_p)
            d.seeds.append(r.seed); d.sample_steps.append(r.issued_samples)
            d.token_ids.append(r.output[-1])
            if r.moved:
                d.table_slots.append(r.slot); d.table_blocks.append(list(r.blocks))
                r.moved = False
            r.issued_ctx += 1
            r.issued_samples += 1
            rows.append(r)
        budget -= len(riders)
        if riders:
            self.stats["rider_tokens"] += len(riders)
        for r in pending:
            if budget <= 0 or len(d.slots) >= self.cfg.max_batch:
                break
            n = min(len(r.context) - r.prefilled, budget)
            last = r.prefilled + n == len(r.context)
            d.slots.append(r.slot); d.q_lens.append(n); d.ctx_lens.append(r.prefilled)
            d.sample.append(1 if last else 0)
            d.temperature.append(r.params.temperature); d.top_k.append(r.params.top_k); d.top_p.append(r.params.top_p)
            d.seeds.append(r.seed); d.sample_steps.append(len(r.output))
            d.token_ids += r.context[r.prefilled:r.prefilled + n]
            if r.prefilled == 0:
                d.table_slots.append(r.slot); d.table_blocks.append(list(r.blocks))
            r.prefilled += n
            r.issued_ctx = r.prefilled
            if last:
                r.issued_samples = len(r.output) + 1
                rows.append(r)
            budget -= n
        self.stats["prefill_steps"] += 1
        self._submit(mb, d, rows)

    def _submit_decode(self, mb: _MicroBatch, replay: bool) -> bool:
        """Issues one decode step of `mb`. False (nothing issued) when a row needs a KV block the
        pool does not have."""
        compact = replay and self._static_decode       # stages replay a captured graph: no per-row fields needed
        if not replay:
            # a new composition: its decode steps cover the rows up to the highest one in use, rounded
            # up to the next size the stages keep a captured step for (rows are handed out lowest first)
            top = max((i + 1 for i, r in enumerate(mb.rows) if r is not None and r.state == "running"), default=0)
            mb.nrows = next((b for b in self._graph_sizes if b >= top), len(mb.rows))
        n_rows = mb.nrows
        if compact:
            d = StepBatch.replay_of(self._next_step_id(), mb.idx, n_rows, 0)
        else:
            d = StepBatch(step_id=self._next_step_id(), micro_batch=mb.idx, replay=replay,
                          token_ids=None if replay else [])
        if not self._grow(mb, d):
            return False
        rows: List[Optional[Request]] = []
        if compact:
            n_live = 0
            for r in mb.rows[:n_rows]:
                live = r is not None and r.state == "running"
                if live:
                    r.issued_ctx += 1
                    r.issued_samples += 1
                    n_live += 1
                rows.append(r if live else None)
            d.live = n_live
            self.stats["decode_steps"] += 1
            self._submit(mb, d, rows)
            return True
        for i, r in enumerate(mb.rows[:n_rows]):
            live = r is not None and r.state == "running"
            # fixed shape: every row of the step is present, dead/empty ones as padding
            d.slots.append(r.slot if r is not None else mb.idx * self.cfg.max_batch + i)
            d.q_lens.append(1 if live else 0)
            d.ctx_lens.append(r.issued_ctx if live else 0)
            d.sample.append(1 if live else 0)
            d.temperature.append(r.params.temperature if live else 0.0)
            d.top_k.append(r.params.top_k if live else 1)
            d.top_p.append(r.params.top_p if live else 1.0)
            d.seeds.append(r.seed if live else 0)
            d.sample_steps.append(r.issued_samples if live else 0)
            if live:
                if r.moved:                      # (only ever set right before a non-replay step)
                    d.table_slots.append(r.slot); d.table_blocks.append(list(r.blocks))
                    r.moved = False
                if not replay:
                    d.token_ids.append(r.output[-1] if r.output else 0)
                r.issued_ctx += 1
                r.issued_samples += 1
            rows.append(r if live else None)
        if not replay and self.head.graph_decode(d):
            # static-buffer decode wants one token per row, padding rows included
            toks, it = [], iter(d.token_ids)
            for q in d.q_lens:
                toks.append(next(it) if q == 1 else 0)
            d.token_ids = toks
        self.stats["decode_steps"] += 1
        self._submit(mb, d, rows)
        return True

    def _dispatch(self, d: StepBatch):
        self.head.control(d)

    def _submit(self, mb: _MicroBatch, d: StepBatch, rows: List[Optional[Request]]):
        d.validate()
        self.stats["steps"] += 1
        mb.inflight.append(_Inflight(d, rows, self.head.submit(d)))

    # ---- results -------------------------------------------------------------------------------
    def _harvest(self, mb: _MicroBatch, res: _Inflight):
        toks = res.get()
        now = time.perf_counter()
        d = res.desc
        per_row = self.head.graph_decode(d)
        k = 0
        for i, r in enumerate(res.rows):
            if per_row:
                tok = toks[i] if i < len(toks) else 0
            else:
                if r is None:
                    continue
                tok = toks[k] if k < len(toks) else 0
                k += 1
            if r is None or r.finished:
                continue
            self._on_token(r, int(tok), now)

    def _on_token(self, r: Request, tok: int, now: float):
        if r.state == "prefill":
            r.state = "running"
            if not r.t_first:               # (a re-prefill after a preemption keeps the first arrival)
                r.t_first = now
                self.ttfts.append(now - r.t_arrival)
                self.hist_ttft.observe(now - r.t_arrival)
        elif r.token_times:
            self.latencies.append(now - r.token_times[-1])
            self.hist_itl.observe(now - r.token_times[-1])
        if r.params.stop_on_eos and tok in r.stop_ids:
            self._finish(r, now)          # EOS is not emitted (orchestration.py:181-183)
            return
        r.output.append(tok)
        r.token_times.append(now)
        self.stats["tokens_out"] += 1
        if r.token_callbacks:               # streaming listeners: they read `r.output` themselves
            for cb in r.token_callbacks:
                try:
                    cb(r)
                except Exception:
                    pass
        if self.cfg.debug_top5 and len(r.output) % 5 == 0:      # progress line (orchestration.py:194-196)
            from ..utils.tracing import get_logger
            get_logger().info("request %d: %d tokens so far: %r", r.rid, len(r.output),
                              self.tok.decode(r.output, skip_special_tokens=True))
        if len(r.output) >= r.params.max_new_tokens or len(r.prompt_ids) + len(r.output) >= self.cfg.max_seq_len:
            self._finish(r, now)

    def _finish(self, r: Request, now: float):
        r.state, r.t_done = "finished", now
        self.stats["completed"] += 1
        if self.tracer is not None:
            self.tracer.instant("scheduler", "finish", {"rid": r.rid, "tokens": len(r.output)})
        r._set_done()

    # ---- introspection -----------------------------------------------------------------------
    def wait_idle(self, timeout_s: float = 5.0) -> bool:
        """Blocks until no request is waiting or running and every finished row has been retired (a
        request is marked done before the scheduler's next tick gives its KV blocks back)."""
        deadline = time.perf_counter() + timeout_s
        while time.perf_counter() < deadline:
            st = self.snapshot()
            if st["waiting"] == 0 and st["running"] == 0 and all(r is None for mb in self.mbs for r in mb.rows):
                return True
            time.sleep(0.002)
        return False

    def snapshot(self) -> Dict:
        lat = sorted(self.latencies)
        p = lambda q: lat[min(len(lat) - 1, int(q * len(lat)))] * 1e3 if lat else None
        return {**self.stats, "waiting": len(self.waiting) + len(self._inbox),
                "running": sum(len(mb.live()) for mb in self.mbs),
                "kv_blocks_used": self.alloc.used, "kv_blocks_total": self.alloc.num_blocks,
                "tpot_ms_p50": p(0.5), "tpot_ms_p99": p(0.99), "fatal": self.fatal,
                "ttft_hist": self.hist_ttft.snapshot(), "itl_hist": self.hist_itl.snapshot()}
