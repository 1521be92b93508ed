"""The scheduler's view of pipeline stage 0.

The scheduler (`Engine`) only ever does three things with the pipeline: submit a step descriptor
and later read the token ids it produced, send a control descriptor (release slots, barrier,
shutdown), and ask how a descriptor will be executed (static-buffer decode or eager). `LocalHead`
does that on a `StageRuntime` living in the scheduler's process — the stage-0 compute, the token
feedback receive and the device-to-host copy of the tokens are enqueued here. `RemoteHead` does it
over two shared-memory rings to a stage-0 process (`head_main`), so that the process that owns the
HTTP server and the scheduler holds no GPU context at all: when a stage hangs inside a device
call, the orchestrator stays responsive, answers with the error envelope and replaces every stage
process. (The reference's orchestrator is such a GPU-less process too, but its failure handling
is an exception string: orchestration.py:118-122,220-228.)
"""
from __future__ import annotations

import queue
import struct
import threading
import time
import weakref
from typing import Dict, List, Optional

import numpy as np
import torch

from ..parallel.transport import Handle
from .batch import KIND_SHUTDOWN, KIND_STEP, StepBatch, control

_RES_MAGIC = 0x52455331      # "RES1"


class StepResult:
    """Token ids of one submitted step, on their way to the scheduler."""

    def __init__(self, tokens=None, event=None, handle: Optional[Handle] = None, host=None):
        self.tokens, self.event, self.handle, self.host = tokens, event, handle, host
        self.t_submit = time.perf_counter()

    def ready(self) -> bool:
        if self.event is not None:
            return self.event.query()
        if self.handle is not None:
            return self.handle.done()
        return True

    def get(self) -> List[int]:
        if self.event is not None:
            self.event.synchronize()
            return self.host.tolist()
        if self.handle is not None:
            self.handle.wait()
        return self.tokens.tolist() if self.tokens is not None else []


class _EventHandle(Handle):
    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)

    def done(self) -> bool:
        return self.event.query()


class _Deferred:
    """A submitted step of an in-process GPU pipeline whose token receive has not been posted yet."""

    __slots__ = ("mb", "n", "buf", "slot", "got", "ev")

    def __init__(self, mb, n, buf, slot):
        self.mb, self.n, self.buf, self.slot = mb, n, buf, slot
        self.got = self.ev = None            # set when the receive is posted


class _DeferredHandle(Handle):
    """Arrival of a deferred step's fed-back tokens: the next step of the micro-batch waits for it on its
    stream, which needs the receive posted first (the only place where the host may wait)."""

    def __init__(self, head, entry):
        # (a weak reference: the head keeps its latest handles, and a cycle through it would leave the whole
        # pipeline — executors, graphs, KV cache — to the cyclic collector, to be released at some later time on
        # whichever thread trips it, instead of when the pipeline is closed)
        self.head, self.entry = weakref.ref(head), entry

    def wait(self):
        head = self.head()
        if head is not None:
            head._pump(upto=self.entry)
        torch.cuda.current_stream().wait_event(self.entry.got)

    def done(self) -> bool:
        head = self.head()
        if head is not None:
            head._pump()
        return self.entry.got is not None and self.entry.got.query()


class _DeferredResult(StepResult):
    def __init__(self, head, entry):
        super().__init__(host=entry.slot[1])
        self.head, self.entry = weakref.ref(head), entry

    def ready(self) -> bool:
        head = self.head()
        if head is not None:
            head._pump()
        return self.entry.ev is not None and self.entry.ev.query()

    def get(self) -> List[int]:
        head = self.head()
        if head is not None:
            head._pump(upto=self.entry)
        self.entry.ev.synchronize()
        return self.host.tolist()


class LocalHead:
    """Stage 0 in this process. Over RCCL hops nothing here blocks the host. Over the in-process device
    hops of a single-process GPU pipeline (tests, rehearsals) a receive can only be posted once the last
    stage's thread has enqueued the matching send: the token receive of a step is therefore posted when the
    scheduler polls the result and finds the message there, not at submission, so submitting never waits.
    The host waits (at most `hop_timeout_s` x stages) only where it has to: a result read before it is ready,
    or the next step of a micro-batch submitted before its fed-back tokens were sent."""

    remote = False

    def __init__(self, runtime):
        self.rt = runtime
        self.links = runtime.links
        self.spec = runtime.ex.spec
        self.num_stages = runtime.links.num_stages
        self.on_gpu = runtime.on_gpu
        # (only where it is used: every stream takes a share of the process's hardware queues)
        self._res_stream = torch.cuda.Stream(device=runtime.device) if self.on_gpu and self.num_stages > 1 else None
        self._last_fb: Dict[int, Handle] = {}        # micro-batch -> arrival of its latest fed-back tokens
        # Token read-back. One stage: the ids go to pinned host memory on the step's own stream, right
        # behind the graph (a side stream for the copy costs more than the copy: work arriving on a
        # second hardware queue once per step slowed every kernel hand-over of a TinyLlama batch-1
        # decode graph, 1.40 against 1.23 ms per token). Several stages: the feedback receive and the
        # host copy run on a side stream, which has to wait for the last stage anyway.
        self._ring: Dict[tuple, list] = {}           # (micro-batch, n) -> [(staging, pinned host, event), ...]
        self._ring_pos: Dict[tuple, int] = {}
        # in-process device hops: steps whose token receive is still to be posted, in submission order
        self._deferred: List[_Deferred] = []
        self._defer = self.on_gpu and self.num_stages > 1 and hasattr(self.links.fb_data, "pending")

    # ---- how a descriptor will run ----------------------------------------------------------
    def graph_decode(self, desc: StepBatch) -> bool:
        return self.rt._graph_decode(desc)

    def result_count(self, desc: StepBatch) -> int:
        return self.rt.result_count(desc)

    def fb_ready(self, mb: int) -> bool:
        """On the GPU the next step waits for the fed-back tokens on its stream; on CPU transports
        that wait would block the calling thread, so a replay step is only submitted once they
        have arrived."""
        h = self._last_fb.get(mb)
        if isinstance(h, _DeferredHandle):
            # in-process device hops: the step would wait on the HOST for the last stage's thread to enqueue
            # the tokens' send, so it is submitted once that has happened (and their receive been posted)
            self._pump()
            return h.entry.got is not None
        if self.on_gpu or self.num_stages == 1 or h is None:
            return True
        return h.done()

    # ---- submission ----------------------------------------------------------------------------
    def control(self, desc: StepBatch):
        self.rt.execute(desc)

    _RING = 6          # > max_inflight_steps + the results the scheduler has not read yet

    def _slot(self, mb: int, n: int):
        """Next pinned host row of micro-batch `mb` for `n` token ids; a slot comes round again only
        after the scheduler has long consumed it, but its event is checked anyway."""
        key = (mb, n)
        ring = self._ring.get(key)
        if ring is None:
            ring = [[None, torch.zeros(max(n, 1), dtype=torch.int32).pin_memory(), None] for _ in range(self._RING)]
            self._ring[key], self._ring_pos[key] = ring, 0
        i = self._ring_pos[key]
        self._ring_pos[key] = (i + 1) % self._RING
        slot = ring[i]
        if slot[2] is not None:
            slot[2].synchronize()
        return slot

    def submit(self, d: StepBatch) -> StepResult:
        rt, mb = self.rt, d.micro_batch
        n_res = rt.result_count(d)
        fb_wait = self._last_fb.get(mb) if d.token_ids is None else None
        with rt.stream_scope(d):           # the stream this step runs on (one per decode micro-batch)
            tokens = rt.execute(d, fb_wait)
            if self.num_stages == 1:
                if not self.on_gpu:
                    return StepResult(tokens=tokens)
                if not n_res:
                    ev = torch.cuda.Event()
                    ev.record()
                    return StepResult(event=ev, host=torch.empty(0, dtype=torch.int32))
                slot = self._slot(mb, n_res)
                host = slot[1]
                host.copy_(tokens, non_blocking=True)      # same stream: done before the next step overwrites `tokens`
                ev = torch.cuda.Event()
                ev.record()
                slot[2] = ev
                return StepResult(event=ev, host=host)
            buf = rt.fb_buffer(mb, n_res)
            if self._defer:
                slot = self._slot(mb, n_res)
                slot[2] = None
                entry = _Deferred(mb, n_res, buf, slot)
                self._deferred.append(entry)
                self._pump()
                self._last_fb[mb] = _DeferredHandle(self, entry)
                return _DeferredResult(self, entry)
            if self.on_gpu:
                slot = self._slot(mb, n_res)
                host = slot[1]
                ev, got = torch.cuda.Event(), torch.cuda.Event()
                with torch.cuda.stream(self._res_stream):
                    h = self.links.fb_data.recv(buf) if n_res else None
                    if h is not None:
                        h.wait()
                    got.record()
                    if n_res:
                        host.copy_(buf, non_blocking=True)
                    ev.record()
                slot[2] = ev
                self._last_fb[mb] = _EventHandle(got)
                return StepResult(event=ev, host=host)
            h = self.links.fb_data.recv(buf) if n_res else None
            self._last_fb[mb] = h
            return StepResult(tokens=buf, handle=h)

    def _pump(self, upto: Optional[_Deferred] = None):
        """Posts the token receives of deferred steps, oldest first, for as long as their messages are there
        (a receive then returns at once). With `upto`: also the ones that are not, up to and including that
        step — the host then waits in `recv` for the last stage's thread, with the hop's deadline."""
        fb = self.links.fb_data
        while self._deferred:
            e = self._deferred[0]
            forced = upto is not None and upto.ev is None
            if e.n and not forced and fb.pending() == 0:
                return
            ev, got = torch.cuda.Event(), torch.cuda.Event()
            with torch.cuda.stream(self._res_stream):
                if e.n:
                    fb.recv(e.buf).wait()
                got.record()
                if e.n:
                    e.slot[1].copy_(e.buf, non_blocking=True)
                ev.record()
            e.slot[2] = ev
            e.got, e.ev = got, ev
            self._deferred.pop(0)

    def shutdown(self):
        self._deferred.clear()
        self._last_fb.clear()
        try:
            if self.links.next_ctrl is not None:
                self.links.next_ctrl.put(control(KIND_SHUTDOWN).to_bytes())
        except Exception:
            pass
        self.rt.drain()


# ---------------------------------------------------------------------------------------------
def pack_result(step_id: int, tokens: List[int], error: str = "") -> bytes:
    err = error.encode()[:2048]
    head = struct.pack("<IiiI", _RES_MAGIC, step_id, len(tokens), len(err))
    return head + np.asarray(tokens, dtype="<i4").tobytes() + err


def unpack_result(buf: bytes):
    magic, step_id, n, nerr = struct.unpack_from("<IiiI", buf, 0)
    if magic != _RES_MAGIC:
        raise ValueError("bad result message")
    off = struct.calcsize("<IiiI")
    toks = np.frombuffer(buf, dtype="<i4", count=n, offset=off).tolist()
    err = bytes(buf[off + 4 * n: off + 4 * n + nerr]).decode(errors="replace")
    return step_id, toks, err


class _RemoteResult:
    def __init__(self, head: "RemoteHead", step_id: int):
        self.head, self.step_id = head, step_id
        self.t_submit = time.perf_counter()

    def ready(self) -> bool:
        self.head._pump()
        return self.step_id in self.head._arrived

    def get(self) -> List[int]:
        t0 = time.perf_counter()
        while not self.ready():
            if time.perf_counter() - t0 > self.head.timeout_s:
                raise TimeoutError(f"no result for step {self.step_id} from stage 0")
            time.sleep(50e-6)
        toks, err = self.head._arrived.pop(self.step_id)
        if err:
            raise RuntimeError(f"stage 0: {err}")
        return toks


class RemoteHead:
    """Stage 0 in another process: descriptors out over `cin`, token ids back over `res`."""

    remote = True
    on_gpu = False          # nothing in this process touches a device

    def __init__(self, spec, num_stages: int, cin, res, graph_capable: bool, graph_rows,
                 debug_top5: bool = False, timeout_s: float = 60.0):
        """`graph_capable` / `graph_rows`: whether the stages run fixed-shape decode steps over static
        buffers (HIP executor) and the row counts of such steps (one, or a list) — the same rule the
        stages apply (`StageRuntime._graph_decode`), needed here to know how many token ids a step returns."""
        self.spec, self.num_stages = spec, num_stages
        self.cin, self.res = cin, res
        self.graph_capable = graph_capable and not debug_top5
        self.graph_rows = frozenset(graph_rows) if isinstance(graph_rows, (list, tuple, set, frozenset)) else frozenset([graph_rows])
        self.timeout_s = timeout_s
        self._arrived: Dict[int, tuple] = {}

    def graph_decode(self, desc: StepBatch) -> bool:
        return self.graph_capable and desc.is_decode and desc.num_seqs in self.graph_rows

    def result_count(self, desc: StepBatch) -> int:
        return desc.num_seqs if self.graph_decode(desc) else desc.num_samples

    def fb_ready(self, mb: int) -> bool:
        return True

    def _pump(self):
        while self.res.pending() > 0:
            msg = self.res.get(timeout_s=0.0)
            if msg is None:
                return
            step_id, toks, err = unpack_result(msg)
            if step_id < 0 and err:        # the stage-0 loop itself failed
                raise RuntimeError(f"stage 0: {err}")
            self._arrived[step_id] = (toks, err)

    def control(self, desc: StepBatch):
        self.cin.put(desc.to_bytes())

    def submit(self, d: StepBatch) -> _RemoteResult:
        self.cin.put(d.to_bytes())
        return _RemoteResult(self, d.step_id)

    def shutdown(self):
        try:
            self.cin.put(control(KIND_SHUTDOWN).to_bytes())
        except Exception:
            pass


def _native_replay_for(runtime, cin, res):
    """The native serving loop of this stage-0 process, when it runs captured decode graphs on a GPU
    over shared-memory rings: the single-stage loop, or — several stages, hops with a native end (RCCL)
    — the stage-0 loop that also forwards the descriptor, sends the hidden states and receives the
    fed-back ids. DLIA_NATIVE_LOOP=0 turns both off, DLIA_NATIVE_HEAD_LOOP=0 the multi-stage one."""
    import os
    if os.environ.get("DLIA_NATIVE_LOOP", "1") == "0":
        return None
    ex, L = runtime.ex, runtime.links
    if not (runtime.on_gpu and hasattr(ex, "decode_step") and getattr(ex, "use_graphs", False)
            and not runtime.debug_top5 and hasattr(cin, "ring") and hasattr(res, "ring")):
        return None
    dev = torch.device(runtime.device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    if L.num_stages == 1:
        from .replay import NativeReplay
        nr = NativeReplay(index, getattr(ex, "decode_states", 1) + 64, cin.ring, res.ring)
    else:
        if os.environ.get("DLIA_NATIVE_HEAD_LOOP", "1") == "0":
            return None
        if not (hasattr(L.next_ctrl, "ring") and hasattr(L.next_data, "native_ends") and hasattr(L.fb_data, "native_ends")):
            return None
        from .replay import NativeHeadLoop
        nr = NativeHeadLoop(index, getattr(ex, "decode_states", 1) + 64, cin.ring, res.ring, L.next_ctrl.ring, L.fb_data, L.next_data)
    nr.set_table(ex.block_table, ex.num_blocks)
    return nr


def _head_main_native(runtime, head, nr, cin, res, idle_beat_s: float) -> int:
    """`head_main` around the native loop: C++ serves replay steps, Python whatever else arrives;
    Python's results join the native queue (one producer for the result ring)."""
    from collections import deque
    from ..parallel.shm import ST_ERROR, ST_READY, ST_STOPPED
    board, ex = runtime.board, runtime.ex
    keep = deque(maxlen=256)             # results (pinned buffers, events) referenced by the native queue
    counted = tokens_counted = busy_counted = 0
    runtime.native_loop = nr
    import os
    if os.environ.get("DLIA_GPU_SPANS", "1") == "0":                 # A/B switch: no device-side step timing
        nr.timeline(False)

    def account():
        nonlocal counted, tokens_counted, busy_counted
        st = nr.stats()
        n = st["native_steps"]
        if n > counted:
            runtime.steps += n - counted
            runtime.fault.count += n - counted
            if board is not None:          # tokens and GPU-busy time (HIP events around the graphs) of the steps served natively
                board.beat(runtime.stage, n - counted, st["tokens"] - tokens_counted, st["gpu_busy_ns"] - busy_counted)
            counted, tokens_counted, busy_counted = n, st["tokens"], st["gpu_busy_ns"]
        if runtime.tracer is not None:
            nr.feed_tracer(runtime.tracer, runtime.stage)

    several = runtime.links.num_stages > 1
    if board is not None:
        board.set_state(runtime.stage, ST_READY)
    step_id = -1
    try:
        while True:
            due = runtime.fault.next_at()                   # a fault rule counts steps: the step it names runs in Python
            nr.handoff_in(due - runtime.fault.count if due is not None else 0)
            if several:
                # what the Python loop left pending on the micro-batch streams becomes a stream dependency:
                # sends out of the static output buffers, the last eager step, the latest fed-back ids
                from ..parallel.pipeline import _enter_native
                _enter_native(runtime)
                for mb, h in list(head._last_fb.items()):
                    if h is not None and runtime._decode_streams:
                        with torch.cuda.stream(runtime._decode_streams[mb % len(runtime._decode_streams)]):
                            h.wait()
                head._last_fb.clear()
            rc, msg = nr.serve(int(idle_beat_s * 1000))
            account()
            if rc == -2:
                break
            if rc == 0:
                if board is not None:
                    board.beat(runtime.stage)
                continue
            desc = StepBatch.from_bytes(msg)
            if desc.kind == KIND_SHUTDOWN:
                nr.flush()
                head.shutdown()
                break
            if desc.kind != KIND_STEP:
                head.control(desc)
                continue
            step_id = desc.step_id
            r = head.submit(desc)
            n = runtime.result_count(desc)
            keep.append(r)
            nr.push_external(step_id, r.event.cuda_event if r.event is not None else None, r.host.data_ptr() if n else 0, n)
            if runtime._graph_decode(desc) and not desc.replay:
                # (re)register the micro-batch's captured step: graph, stream, where its ids land
                st = ex._mb_state(desc.micro_batch, desc.num_seqs)
                if st.graph is not None and several:
                    nr.set_mb(desc.micro_batch, st.graph.raw_cuda_graph_exec(), runtime._stream_of(desc).cuda_stream,
                              ex.input_buffer(desc.micro_batch, desc.num_seqs), st.hidden_out, desc.num_seqs)
                elif st.graph is not None:
                    nr.set_mb(desc.micro_batch, st.graph.raw_cuda_graph_exec(), runtime._stream_of(desc).cuda_stream,
                              st.tokens.data_ptr(), desc.num_seqs)
    except BaseException as e:
        if board is not None:
            board.set_state(runtime.stage, ST_ERROR, f"{type(e).__name__}: {e}")
        try:
            res.put(pack_result(step_id if isinstance(e, Exception) else -1, [], f"{type(e).__name__}: {e}"))
            res.put(pack_result(-1, [], f"{type(e).__name__}: {e}"))
        except Exception:
            pass
        runtime.abort()
        raise
    finally:
        try:
            import json
            import os
            path = os.environ.get("DLIA_NATIVE_LOOP_STATS")        # tests / benches: how many steps never entered Python
            if path:
                with open(path, "w") as f:
                    json.dump(nr.stats(), f)
            runtime.native_stats = nr.stats()
            runtime.dump_trace()
            runtime.native_loop = None
            nr.close()
        except Exception:
            pass
    if board is not None:
        board.set_state(runtime.stage, ST_STOPPED)
    return runtime.steps


def head_main(runtime, cin, res, idle_beat_s: float = 0.5) -> int:
    """Loop of a stage-0 process serving a `RemoteHead`: descriptors in, token ids out, until
    SHUTDOWN. A forwarder thread polls the submitted steps' tokens (device event / feedback receive)
    and sends each result when it is there, so the descriptor loop itself never blocks on the device. A single-stage GPU
    pipeline runs the loop natively instead (`runtime/replay.py`)."""
    from ..parallel.shm import ST_ERROR, ST_READY, ST_STOPPED, RingClosed
    head = LocalHead(runtime)
    nr = _native_replay_for(runtime, cin, res)
    if nr is not None:
        return _head_main_native(runtime, head, nr, cin, res, idle_beat_s)
    board = runtime.board
    pending: "queue.Queue" = queue.Queue()
    failed = threading.Event()
    res_lock = threading.Lock()          # the ring is single-producer: forwarder and error path take turns

    def reply(msg: bytes):
        with res_lock:
            res.put(msg)

    def forward():
        # Results leave as their steps complete (the scheduler matches them by step id): a decode
        # step's tokens do not wait behind an earlier, longer prefill chunk of another micro-batch.
        waiting = []
        closing = False
        if runtime.on_gpu and torch.device(runtime.device).index is not None:
            torch.cuda.set_device(torch.device(runtime.device))      # (HIP's current device is per thread)
        while True:
            try:
                item = pending.get(timeout=None if not waiting else 50e-6)
                if item is None:
                    closing = True
                else:
                    waiting.append(item)
                    continue                      # take everything that is queued before polling
            except queue.Empty:
                pass
            still = []
            for step_id, r in waiting:
                try:
                    if not closing and not r.ready():
                        still.append((step_id, r))
                        continue
                    reply(pack_result(step_id, r.get()))
                except BaseException as e:          # reported to the scheduler, which fails the requests
                    failed.set()
                    if board is not None:
                        board.set_state(runtime.stage, ST_ERROR, f"{type(e).__name__}: {e}")
                    try:
                        reply(pack_result(step_id, [], f"{type(e).__name__}: {e}"))
                    except Exception:
                        pass
                    return
            waiting = still
            if closing:
                return

    th = threading.Thread(target=forward, name="dlia-head-results", daemon=True)
    th.start()
    if board is not None:
        board.set_state(runtime.stage, ST_READY)
    try:
        while not failed.is_set():
            try:
                msg = cin.get(timeout_s=idle_beat_s)
            except RingClosed:
                break
            if msg is None:
                if board is not None:
                    board.beat(runtime.stage)
                continue
            desc = StepBatch.from_bytes(msg)
            if desc.kind == KIND_SHUTDOWN:
                head.shutdown()
                break
            if desc.kind == KIND_STEP:
                pending.put((desc.step_id, head.submit(desc)))
            else:
                head.control(desc)
    except BaseException as e:
        if board is not None:
            board.set_state(runtime.stage, ST_ERROR, f"{type(e).__name__}: {e}")
        try:
            reply(pack_result(-1, [], f"{type(e).__name__}: {e}"))
        except Exception:
            pass
        runtime.abort()
        raise
    finally:
        pending.put(None)
        th.join(5.0)
        try:
            runtime.dump_trace()
        except Exception:
            pass
    if board is not None and not failed.is_set():
        board.set_state(runtime.stage, ST_STOPPED)
    return runtime.steps
