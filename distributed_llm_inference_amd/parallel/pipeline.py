"""Per-stage pipeline runtime.

`StageRuntime.execute(desc)` is what one stage does for one step descriptor:

    stage 0      : tokens (explicit, or the ids the last stage fed back for this micro-batch)
                   -> embed -> layers -> forward descriptor, send hidden to stage 1
    stage i      : recv hidden -> layers -> forward descriptor, send hidden
    last stage   : recv hidden -> layers -> final norm + lm_head on the sampled rows -> sampler
                   -> send int32 token ids back to stage 0
    (one stage   : all of the above locally)

Decode steps run over the micro-batch's static buffers (hipGraph replay on the GPU backend):
receives land directly in the graph's input buffer and sends leave from its output buffer.
All GPU work and RCCL transfers are enqueued asynchronously; the host only ever blocks on the
control ring waiting for the next descriptor.

`stage_main` is the loop every non-zero rank runs; rank 0 drives its own StageRuntime from the
engine's scheduler. This replaces the reference's Flask `/process` worker (Worker1.py:208-245)
and the orchestrator's relay loop (orchestration.py:109-137) with a true stage->stage chain.
"""
from __future__ import annotations

import collections
import contextlib
import os
import time
import traceback
from typing import Deque, Dict, List, Optional

import torch

from ..runtime.batch import KIND_BARRIER, KIND_RELEASE, KIND_SHUTDOWN, KIND_STEP, StepBatch
from .transport import Handle, PipelineLinks


class FaultInjector:
    """Test hook: "stage1:exit@3" (die at step 3), "stage1:stall@2:5.0" (sleep 5 s at step 2),
    "stage0:raise@1" (raise inside execute). Several rules separated by commas."""

    def __init__(self, spec: str, stage: int):
        self.rules = []
        for part in (spec or "").split(","):
            part = part.strip()
            if not part:
                continue
            who, what = part.split(":", 1)
            if who != f"stage{stage}":
                continue
            action, rest = what.split("@", 1)
            bits = rest.split(":")
            self.rules.append((action, int(bits[0]), float(bits[1]) if len(bits) > 1 else 0.0))
        self.count = 0

    def next_at(self) -> Optional[int]:
        """Step count of the next rule still ahead (None: no rule left). The native loops serve
        steps without entering Python; they hand exactly that step up so that it is counted here."""
        ahead = [at for _, at, _ in self.rules if at > self.count]
        return min(ahead) if ahead else None

    def tick(self):
        self.count += 1
        for action, at, arg in self.rules:
            if self.count != at:
                continue
            if action == "exit":
                os._exit(17)
            if action == "stall":
                time.sleep(arg)
            if action == "raise":
                raise RuntimeError(f"injected fault at step {at}")


class GpuSpans:
    """Device-side timing of the steps the Python loop executes: HIP timing events recorded around a
    step's wait for its input, its kernels and its send; completed steps become spans on the host
    clock (Chrome trace) and the compute spans add up — as a union over the streams — to the stage's
    GPU-busy time, which `/metrics` reports. The native loops keep the same books for the steps they
    serve (`ops/csrc/comm/replay_loop.cpp`)."""

    def __init__(self, device):
        self.device = device
        self._open: Deque[tuple] = collections.deque()
        self._anchor = None                    # (event, host ns)
        self._pool: List["torch.cuda.Event"] = []
        self.busy_ns = 0
        self._busy_until_us = 0.0

    def _anchored(self):
        """The device-clock anchor: an event stamped with the host clock at its completion, recorded on the CURRENT
        stream and waited for. Not on a stream of its own: one more stream in the process made two micro-batch
        streams share a hardware queue (HIP maps streams onto GPU_MAX_HW_QUEUES queues, 4 by default), which cost
        9-13 % of the throughput of 256-row steps; the timing events themselves cost nothing measurable. Renewed
        every few minutes (float milliseconds between events), at the price of waiting for what the stream has queued."""
        now = time.perf_counter_ns()
        if self._anchor is None or now - self._anchor[1] > 300e9:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            ev.synchronize()
            self._anchor = (ev, time.perf_counter_ns())
        return self._anchor

    def stamp(self, stream=None, pooled: bool = False) -> "torch.cuda.Event":
        """`pooled`: the event will bound ONE span added with `recycle=True` (adjacent spans of a traced step share
        their boundary events, those are not pooled)."""
        ev = self._pool.pop() if pooled and self._pool else torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream(self.device))
        return ev

    def add(self, track: str, name: str, e0, e1, args: Optional[dict] = None, busy: bool = False, recycle: bool = False):
        self._open.append((self._anchored(), track, name, e0, e1, args or {}, busy, recycle))

    def reap(self, tracer=None, wait: bool = False) -> int:
        """Resolves finished spans (all of them with `wait`); returns the GPU-busy ns they added."""
        added = 0
        while self._open:
            (aev, ans), track, name, e0, e1, args, busy, recycle = self._open[0]
            if wait:
                e1.synchronize()
            elif not e1.query():
                break
            self._open.popleft()
            t0 = ans / 1e3 + aev.elapsed_time(e0) * 1e3
            t1 = ans / 1e3 + aev.elapsed_time(e1) * 1e3
            if busy and t1 > t0:
                b = max(t0, self._busy_until_us)
                if t1 > b:
                    added += int((t1 - b) * 1e3)
                    self._busy_until_us = t1
            if tracer is not None:
                tracer.span(track, name, int(t0 * 1e3), int(max(t0, t1) * 1e3), args)
            if recycle and len(self._pool) < 64:
                self._pool += [e0, e1]
        self.busy_ns += added
        return added


class StageRuntime:
    def __init__(self, executor, links: PipelineLinks, board=None, fault: str = "", tracer=None,
                 trace_path: Optional[str] = None, decode_streams: int = 3, debug_top5: bool = False):
        self.ex = executor
        self.links = links
        self.board = board
        self.stage = links.stage
        self.fault = FaultInjector(fault, links.stage)
        self.tracer = tracer
        self.trace_path = trace_path
        self.device = getattr(executor, "device", torch.device("cpu"))
        self.on_gpu = torch.device(self.device).type == "cuda"
        self.hidden = executor.spec.hidden_size
        self.act_dtype = torch.bfloat16 if self.on_gpu else getattr(executor, "dtype", torch.float32)
        self._send_pending: Dict[int, Handle] = {}     # micro-batch -> last send of its output buffer
        # Sends out of eager steps leave from freshly allocated tensors. The caching allocator hands a
        # freed block back to the stream it was allocated on (the eager stream) at once, so such a tensor
        # is kept here until THAT stream has waited for its send: the next eager step does so first thing.
        self._eager_sends: List[tuple] = []
        self._fb_bufs: Dict[tuple, torch.Tensor] = {}
        self._eager_in: Optional[torch.Tensor] = None
        self._local_fb: Dict[int, torch.Tensor] = {}   # single-stage pipeline: last tokens per micro-batch
        self.steps = 0
        self.barrier_times: List[float] = []
        self.debug_top5 = debug_top5      # log the 5 most likely tokens of a request's first 3 draws
        self.top5_log: List[dict] = []
        # Decode micro-batches own their workspaces, so consecutive micro-batches go to different
        # HIP streams: the ramp-up and tail of one micro-batch's kernels (and its recv/send) overlap
        # the other's instead of leaving the chip part-idle at every kernel boundary. Eager steps
        # (prefill) share one workspace and stay on the main stream.
        self._main_stream = torch.cuda.current_stream(self.device) if self.on_gpu else None
        n = max(1, int(decode_streams)) if self.on_gpu and hasattr(executor, "decode_step") else 0
        self._decode_streams = [torch.cuda.Stream(device=self.device) for _ in range(n)]
        # micro-batch -> event after its latest eager step: only that micro-batch's decode steps wait for it
        self._eager_done: Dict[int, "torch.cuda.Event"] = {}
        self._eager_barrier = os.environ.get("DLIA_EAGER_BARRIER") == "1"    # A/B: eager steps order against every micro-batch
        # (DLIA_GPU_SPANS=0: no device-side step timing at all — an A/B switch)
        self.spans = GpuSpans(self.device) if self.on_gpu and os.environ.get("DLIA_GPU_SPANS", "1") != "0" else None
        self._probe_stream = None         # tracing only: observes when a send has left without holding the compute stream up

    # ---- helpers ---------------------------------------------------------------------------
    def _graph_decode(self, desc: StepBatch) -> bool:
        """Static-buffer (hipGraph) decode: single-token steps over a whole fixed-shape micro-batch.
        Anything else — a prefill chunk, also one made only of 1-token prompts — runs eagerly and
        leaves the micro-batch's captured state alone."""
        ex = self.ex
        return (desc.is_decode and hasattr(ex, "decode_step") and not self.debug_top5
                and desc.num_seqs in getattr(ex, "graph_sizes", (desc.num_seqs,)))

    def result_count(self, desc: StepBatch) -> int:
        """Number of token ids the last stage produces for `desc`: a static-buffer decode step
        yields one per row (padding rows included), an eager step one per sampled sequence."""
        return desc.num_seqs if self._graph_decode(desc) else desc.num_samples

    def fb_buffer(self, mb: int, n: int) -> torch.Tensor:
        """Stage 0: where the fed-back tokens of micro-batch `mb` land."""
        if hasattr(self.ex, "input_buffer") and self.ex.plan.has_embed and n in getattr(self.ex, "graph_sizes", ()):
            return self.ex.input_buffer(mb, n)      # the static token buffer the decode graph reads
        key = (mb, n)
        b = self._fb_bufs.get(key)
        if b is None:
            b = torch.zeros(n, dtype=torch.int32, device=self.device)
            self._fb_bufs[key] = b
        return b

    # ---- streams ----------------------------------------------------------------------------
    def _stream_of(self, desc: StepBatch):
        if not self._decode_streams or desc.kind != KIND_STEP or not self._graph_decode(desc):
            return self._main_stream
        return self._decode_streams[desc.micro_batch % len(self._decode_streams)]

    def stream_scope(self, desc: StepBatch):
        """Context in which `execute(desc)` enqueues its device work; callers that post-process the
        result (copy tokens to the host, record events) do it inside the same scope."""
        s = self._stream_of(desc)
        return torch.cuda.stream(s) if s is not None else contextlib.nullcontext()

    # ---- one descriptor ----------------------------------------------------------------------
    def execute(self, desc: StepBatch, fb_wait: Optional[Handle] = None) -> Optional[torch.Tensor]:
        """Runs this stage's share of `desc`. Returns sampled tokens (int32 tensor, device) when
        this is the last stage, else None. `fb_wait`: completion of the feedback receive that
        filled the stage-0 token buffer for this step."""
        if desc.kind != KIND_STEP or not self.on_gpu:
            return self._execute(desc, fb_wait)
        s = self._stream_of(desc)
        mb = desc.micro_batch
        with torch.cuda.stream(s):
            if s is self._main_stream:
                # An eager step (a prefill chunk) shares nothing with the decode steps of OTHER micro-batches:
                # it has its own workspace, writes the block-table rows and KV blocks of its own sequences
                # only, and eager steps are ordered among themselves by this stream. So it waits for the
                # decode steps already queued for its own micro-batch and nothing else; the graphs of the
                # other micro-batches keep replaying next to it, and a chunk no longer stops every stream
                # of the stage for its whole duration.
                if self._eager_barrier:
                    for ds in self._decode_streams:       # A/B switch: the chunk stops every stream of the stage
                        s.wait_stream(ds)
                elif self._decode_streams:
                    s.wait_stream(self._decode_streams[mb % len(self._decode_streams)])
                out = self._execute(desc, fb_wait)
                ev = torch.cuda.Event()
                ev.record(s)
                if self._eager_barrier:
                    self._eager_done = {m: ev for m in range(max(len(self._decode_streams), getattr(self.ex, "decode_states", 1)))}
                else:
                    self._eager_done[mb] = ev
            else:
                ev = self._eager_done.pop(mb, None)
                if ev is not None:
                    s.wait_event(ev)                      # prefill of these sequences has run (later steps: stream order)
                out = self._execute(desc, fb_wait)
        return out

    def _execute(self, desc: StepBatch, fb_wait: Optional[Handle] = None) -> Optional[torch.Tensor]:
        L = self.links
        if desc.kind != KIND_STEP:
            if desc.kind == KIND_RELEASE:
                if self.on_gpu:
                    for ds in self._decode_streams:       # steps still reading these rows finish first
                        torch.cuda.current_stream(self.device).wait_stream(ds)
                self.ex.release(desc.slots)
            if L.next_ctrl is not None:
                L.next_ctrl.put(desc.to_bytes())
            if desc.kind == KIND_BARRIER:
                self.barrier()
            return None
        t_start = time.perf_counter_ns()
        self.fault.tick()
        mb, B, T = desc.micro_batch, desc.num_seqs, desc.num_tokens
        graph = self._graph_decode(desc)
        sp, traced = self.spans, self.tracer is not None and self.spans is not None
        span_args = {"mb": mb, "tokens": T, "decode": desc.is_decode, "step": desc.step_id, "loop": "python"}
        e_free = sp.stamp() if traced else None        # the stream is free for this step from here on
        if not graph and self._eager_sends:
            for h, _keep in self._eager_sends:         # (stream-side waits on transfers that left long ago)
                h.wait()
            self._eager_sends.clear()

        # ---- input ----------------------------------------------------------------------
        x = None
        if L.is_first:
            if desc.token_ids is not None:
                x = torch.tensor(desc.token_ids, dtype=torch.int32)
                if graph:
                    x = None          # decode_step uploads explicit ids itself (non-replay path)
            else:
                if fb_wait is not None:
                    fb_wait.wait()
                if graph:
                    # tokens are already in the graph's static input — except right after an eager
                    # step of a single-stage pipeline, whose tokens are still in a loose tensor
                    x = None
                    if L.num_stages == 1 and not desc.replay:
                        prev = self._local_fb.get(mb)
                        if prev is not None and prev.numel() == B:
                            x = prev
                            if self.on_gpu:           # allocated by the eager stream, read by this one
                                prev.record_stream(torch.cuda.current_stream(self.device))
                elif L.num_stages == 1:
                    x = self._local_fb[mb]
                else:
                    x = self.fb_buffer(mb, T)
        else:
            if graph:
                buf = self.ex.input_buffer(mb, B)
            else:
                if self._eager_in is None or self._eager_in.shape[0] < T:
                    self._eager_in = torch.zeros(max(T, 256), self.hidden, dtype=self.act_dtype, device=self.device)
                buf = self._eager_in[:T]
            L.prev_data.recv(buf).wait()
            x = None if graph else buf

        # ---- compute ---------------------------------------------------------------------
        prev_send = self._send_pending.pop(mb, None)
        if prev_send is not None:
            prev_send.wait()          # the output buffer of this micro-batch is free again
        e_in = sp.stamp(pooled=not traced) if sp is not None else None   # input there, output buffer free: compute starts
        if traced:
            sp.add(f"stage{self.stage}.gpu.mb{mb}", "recv-wait", e_free, e_in, span_args)
        if graph:
            out = self.ex.decode_step(desc, x)
        else:
            out = self.ex.forward(desc, x)
            if L.is_last:
                if self.debug_top5:
                    self._log_top5(desc, out)
                out = self.ex.sample(desc, out)
            elif L.num_stages > 1:
                out = out.clone()     # eager outputs alias a workspace the next step overwrites

        e_out = None
        if sp is not None:
            e_out = sp.stamp(pooled=not traced)
            sp.add(f"stage{self.stage}.gpu.mb{mb}", "compute", e_in, e_out, span_args, busy=True, recycle=not traced)

        # ---- output ----------------------------------------------------------------------
        result = None
        if L.is_last:
            result = out
            if L.num_stages == 1:
                self._local_fb[mb] = out
            elif out.numel() > 0:
                self._send_pending[mb] = L.fb_data.send(out)
                if not graph and self.on_gpu:
                    self._eager_sends.append((self._send_pending[mb], out))
        else:
            L.next_ctrl.put(desc.to_bytes())           # descriptor first: the next stage posts its recv
            self._send_pending[mb] = L.next_data.send(out)
            if not graph and self.on_gpu:
                self._eager_sends.append((self._send_pending[mb], out))
        if traced and mb in self._send_pending and L.num_stages > 1:
            # when the send has left: observed from a side stream, the compute stream is not held up
            if self._probe_stream is None:
                self._probe_stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._probe_stream):
                try:
                    self._send_pending[mb].wait()
                    sp.add(f"stage{self.stage}.hop-out", "send", e_out, sp.stamp(), span_args)
                except Exception:
                    pass
        self.steps += 1
        if self.board is not None:
            # busy = GPU time of the steps that have completed since the last beat (HIP events), not host time
            self.board.beat(self.stage, 1, T, sp.reap(self.tracer) if sp is not None else time.perf_counter_ns() - t_start)
        if self.tracer is not None:
            self.tracer.span(f"stage{self.stage}.host", "enqueue step", t_start, time.perf_counter_ns(),
                             {"mb": mb, "tokens": T, "decode": desc.is_decode})
        return result

    def _log_top5(self, desc: StepBatch, logits: torch.Tensor):
        """The reference prints the top-5 candidates of the first three generation steps
        (orchestration.py:172-178). Here: logged per sampled row whose draw counter is < 3, from
        the logits of an eager step (the debug flag turns graph capture off for that reason)."""
        from ..runtime.sampler import top_candidates
        from ..utils.tracing import get_logger
        rows = [i for i in range(desc.num_seqs) if desc.sample[i] and desc.q_lens[i] > 0]
        lg = logits.detach().float().cpu()
        for k, i in enumerate(rows):
            if desc.sample_steps[i] >= 3 or k >= lg.shape[0]:
                continue
            probs, ids = top_candidates(lg[k], desc.temperature[i], desc.top_k[i], desc.top_p[i], 5)
            rec = {"slot": desc.slots[i], "draw": desc.sample_steps[i],
                   "top5": [(int(t), round(float(pr), 4)) for pr, t in zip(probs, ids)]}
            if len(self.top5_log) < 4096:
                self.top5_log.append(rec)
            get_logger().info("top-5 slot %d draw %d: %s", rec["slot"], rec["draw"],
                              ", ".join(f"{t} ({pr:.1%})" for t, pr in rec["top5"]))

    def barrier(self):
        """Finish everything this stage has enqueued, then meet the other stages. Records the
        host time right after the barrier (used to bracket timed benchmark regions)."""
        self.drain()
        if self.links.num_stages > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                if self.on_gpu and dist.get_backend() == "nccl":
                    dist.barrier(device_ids=[torch.device(self.device).index or 0])
                else:
                    dist.barrier()
        if self.on_gpu:
            torch.cuda.synchronize(self.device)
        self.barrier_times.append(time.perf_counter())

    def dump_trace(self):
        """Stage 0 writes `trace_path` and folds in the files the other stages left next to it."""
        if self.tracer is None or not self.trace_path:
            return None
        if self.spans is not None:
            self.spans.reap(self.tracer, wait=True)
        nl = getattr(self, "native_loop", None)
        if nl is not None and getattr(nl, "h", None):
            nl.feed_tracer(self.tracer, self.stage)
        from ..utils.tracing import stage_trace_path
        path = stage_trace_path(self.trace_path, self.stage)
        return self.tracer.dump(path, merge_glob=(self.trace_path + ".stage*") if self.stage == 0 else None)

    def drain(self, timeout_s: Optional[float] = None):
        """Finish what this stage has enqueued. With a deadline the host polls instead of parking in
        a device synchronise, and a stage that does not come idle in time (a transfer spinning on a
        peer that died) aborts its hops and raises: nothing here can hang forever."""
        for h in self._send_pending.values():
            try:
                h.wait()
            except Exception:
                pass
        self._send_pending.clear()
        if not self.on_gpu:
            return
        if timeout_s is None:
            torch.cuda.synchronize(self.device)
            return
        evs = []
        for s in [self._main_stream, *self._decode_streams]:
            ev = torch.cuda.Event()
            ev.record(s)
            evs.append(ev)
        t0 = time.perf_counter()
        while not all(ev.query() for ev in evs):
            if time.perf_counter() - t0 > timeout_s:
                self.abort()
                raise TimeoutError(f"stage {self.stage}: device work did not finish within {timeout_s:.1f}s")
            time.sleep(0.001)

    def abort(self):
        """Abandon every hop of this stage (RCCL: ncclCommAbort) without waiting for the device."""
        self._send_pending.clear()
        self._eager_sends.clear()
        self.links.abort()


def _native_stage_loop_for(runtime: StageRuntime):
    """The native loop of a stage behind stage 0 (`runtime/replay.py::NativeStageLoop`), when the stage
    runs captured decode graphs on a GPU, its descriptors travel over shared-memory rings and its
    hops have a native end (in-process device hops, RCCL hops). DLIA_NATIVE_LOOP=0 turns it off, and
    DLIA_NATIVE_STAGE_LOOP=0 for RCCL hops only."""
    if os.environ.get("DLIA_NATIVE_LOOP", "1") == "0":
        return None
    L, ex = runtime.links, runtime.ex
    edge_out = L.fb_data if L.is_last else L.next_data
    if not (runtime.on_gpu and L.num_stages > 1 and not L.is_first and hasattr(ex, "decode_step") and getattr(ex, "use_graphs", False)
            and not runtime.debug_top5
            and hasattr(L.prev_ctrl, "ring") and (L.next_ctrl is None or hasattr(L.next_ctrl, "ring"))
            and hasattr(L.prev_data, "native_ends") and hasattr(edge_out, "native_ends")):
        return None
    if type(L.prev_data).__name__ == "RcclEdge" and os.environ.get("DLIA_NATIVE_STAGE_LOOP", "1") == "0":
        return None
    from ..runtime.replay import NativeStageLoop
    dev = torch.device(runtime.device)
    nl = NativeStageLoop(dev.index if dev.index is not None else torch.cuda.current_device(), getattr(ex, "decode_states", 1) + 64,
                         L.prev_ctrl.ring, L.next_ctrl.ring if L.next_ctrl is not None else None, L.prev_data, edge_out)
    nl.set_table(ex.block_table, ex.num_blocks)
    return nl


def _enter_native(runtime: StageRuntime):
    """Hand the micro-batch streams over to the native loop: whatever the Python loop left pending on
    them — the last eager step, sends out of the static output buffers — becomes a stream dependency."""
    for mb, h in list(runtime._send_pending.items()):
        s = runtime._decode_streams[mb % len(runtime._decode_streams)]
        with torch.cuda.stream(s):
            h.wait()
    runtime._send_pending.clear()
    for mb, ev in list(runtime._eager_done.items()):
        runtime._decode_streams[mb % len(runtime._decode_streams)].wait_event(ev)
    runtime._eager_done.clear()


def stage_main(runtime: StageRuntime, idle_beat_s: float = 0.5, drain_timeout_s: float = 30.0) -> int:
    """Blocking loop of a non-zero stage: descriptors in, steps out, until SHUTDOWN.
    Returns the number of steps executed. Errors are recorded on the status board and re-raised.
    Replay steps of captured micro-batches are served by the native loop where there is one
    (`_native_stage_loop_for`); everything else — prefill chunks, recompositions, control messages —
    runs here."""
    from .shm import ST_ERROR, ST_READY, ST_STOPPED, RingClosed
    L, board = runtime.links, runtime.board
    if board is not None:
        board.set_state(runtime.stage, ST_READY)
    nl = None
    counted = tokens_counted = busy_counted = 0
    try:
        nl = _native_stage_loop_for(runtime)
        runtime.native_loop = nl
        if nl is not None and os.environ.get("DLIA_GPU_SPANS", "1") == "0":
            nl.timeline(False)
        while True:
            if nl is not None:
                _enter_native(runtime)
                due = runtime.fault.next_at()               # a fault rule counts steps: the step it names runs in Python
                nl.handoff_in(due - runtime.fault.count if due is not None else 0)
                rc, msg = nl.serve(int(idle_beat_s * 1000))
                st = nl.stats()
                n = st["native_steps"]
                if n > counted:
                    runtime.steps += n - counted
                    runtime.fault.count += n - counted
                    runtime.ex.graph_replays += n - counted
                    if board is not None:
                        board.beat(runtime.stage, n - counted, st["tokens"] - tokens_counted, st["gpu_busy_ns"] - busy_counted)
                    counted, tokens_counted, busy_counted = n, st["tokens"], st["gpu_busy_ns"]
                if runtime.tracer is not None:
                    nl.feed_tracer(runtime.tracer, runtime.stage)
                if rc == -2:
                    break
                if rc == 0:
                    msg = None
            else:
                try:
                    msg = L.prev_ctrl.get(timeout_s=idle_beat_s)
                except RingClosed:
                    break
            if msg is None:
                if board is not None:
                    board.beat(runtime.stage)          # idle heartbeat
                continue
            desc = StepBatch.from_bytes(msg)
            if desc.kind == KIND_SHUTDOWN:
                if L.next_ctrl is not None:
                    L.next_ctrl.put(msg)
                break
            runtime.execute(desc)
            if nl is not None and desc.kind == KIND_STEP and runtime._graph_decode(desc) and not desc.replay:
                # (re)register the micro-batch's captured step: graph, stream, static input and output
                st = runtime.ex._mb_state(desc.micro_batch, desc.num_seqs)
                if st.graph is not None:
                    nl.set_mb(desc.micro_batch, st.graph.raw_cuda_graph_exec(), runtime._stream_of(desc).cuda_stream,
                              st.hidden_in, st.tokens if L.is_last else st.hidden_out, desc.num_seqs)
    except BaseException as e:
        if board is not None:
            board.set_state(runtime.stage, ST_ERROR, f"{type(e).__name__}: {e}")
        traceback.print_exc()
        runtime.abort()                 # a failed stage never waits for its neighbours again
        raise
    finally:
        try:
            runtime.drain(timeout_s=drain_timeout_s)
            runtime.dump_trace()
        except Exception:
            pass
        if nl is not None:
            runtime.native_stats = nl.stats()
            runtime.native_loop = None
            nl.close()
    if board is not None:
        board.set_state(runtime.stage, ST_STOPPED)
    return runtime.steps
