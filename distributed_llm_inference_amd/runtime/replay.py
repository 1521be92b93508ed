"""ctypes front of the native serving loop (`ops/csrc/comm/replay_loop.cpp` -> `ops/_comm.so`).

`NativeReplay.serve()` blocks inside C++ (the GIL is released) and handles the steady state of a
single-stage pipeline — compact replay descriptors in, hipGraph launch, token ids out — returning to
Python only for messages that need it. See `runtime/head.py::head_main`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

_sigs_done = False


def _lib():
    global _sigs_done
    from ..parallel import rccl
    L = rccl.lib()
    if not _sigs_done:
        vp, ci, u32 = C.c_void_p, C.c_int, C.c_uint32
        sig = {
            "dlia_replay_create": (vp, [ci, ci, vp, vp]),
            "dlia_replay_error": (C.c_char_p, [vp]),
            "dlia_replay_set_table": (ci, [vp, vp, ci, ci, ci]),
            "dlia_replay_set_mb": (ci, [vp, ci, vp, vp, vp, ci]),
            "dlia_replay_push_external": (ci, [vp, ci, vp, vp, ci]),
            "dlia_replay_pending": (ci, [vp]),
            "dlia_replay_stats": (None, [vp, C.POINTER(C.c_uint64)]),
            "dlia_replay_serve": (ci, [vp, vp, vp, vp, u32, C.POINTER(u32), ci]),
            "dlia_replay_flush": (ci, [vp, vp, ci]),
            "dlia_replay_destroy": (None, [vp]),
            "dlia_stage_set_edges": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
            "dlia_replay_timeline": (None, [vp, ci]),
            "dlia_replay_read_spans": (ci, [vp, C.POINTER(C.c_double), ci]),
            "dlia_replay_handoff_in": (None, [vp, C.c_long]),
            "dlia_stage_set_mb": (ci, [vp, ci, vp, vp, vp, C.c_size_t, vp, C.c_size_t, ci]),
            "dlia_stage_serve": (ci, [vp, vp, vp, vp, u32, C.POINTER(u32), ci]),
            "dlia_head_set_out": (ci, [vp, ci, vp, C.c_size_t]),
            "dlia_head_serve": (ci, [vp, vp, vp, vp, vp, u32, C.POINTER(u32), ci]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _sigs_done = True
    return L


class NativeReplayError(RuntimeError):
    pass


SPAN_FIELDS = ("step_id", "mb", "rows", "host0_us", "host1_us", "free_us", "compute0_us", "compute1_us", "send1_us")


class _LoopCommon:
    """What both native loops share: counters, the step timeline, the fault-injection hand-off."""

    def stats(self):
        out = (C.c_uint64 * 8)()
        self.L.dlia_replay_stats(self.h, out)
        return {"native_steps": int(out[0]), "handed_to_python": int(out[1]), "gpu_busy_ns": int(out[2]),
                "recv_wait_ns": int(out[3]), "send_ns": int(out[4]), "spans_dropped": int(out[5]), "tokens": int(out[6]),
                "spans_waiting": int(out[7])}

    def timeline(self, on: bool):
        self.L.dlia_replay_timeline(self.h, 1 if on else 0)

    def read_spans(self, max_spans: int = 4096):
        """Finished steps of this loop as dictionaries (`SPAN_FIELDS`; times in microseconds of the
        host's monotonic clock — the clock `time.perf_counter_ns` reads; -1 where a stamp does not
        apply). Each span is returned once."""
        n_f = len(SPAN_FIELDS)
        buf = (C.c_double * (max_spans * n_f))()
        n = self.L.dlia_replay_read_spans(self.h, buf, max_spans)
        return [dict(zip(SPAN_FIELDS, buf[i * n_f:(i + 1) * n_f])) for i in range(n)]

    def handoff_in(self, n: int):
        """The n-th replay step from now goes to Python instead of being served natively (n <= 0: never)."""
        self.L.dlia_replay_handoff_in(self.h, int(n))

    def feed_tracer(self, tracer, stage: int):
        """Moves the finished spans into a Chrome-trace `Tracer`: enqueue (host), recv-wait, compute, send."""
        if tracer is None:
            return
        for sp in self.read_spans():
            args = {"mb": int(sp["mb"]), "rows": int(sp["rows"]), "step": int(sp["step_id"]), "loop": "native"}
            ns = lambda k: int(sp[k] * 1e3)
            if sp["host1_us"] >= sp["host0_us"] > 0:
                tracer.span(f"stage{stage}.host", "enqueue step", ns("host0_us"), ns("host1_us"), args)
            if sp["free_us"] >= 0 and sp["compute0_us"] >= sp["free_us"]:
                tracer.span(f"stage{stage}.gpu.mb{int(sp['mb'])}", "recv-wait", ns("free_us"), ns("compute0_us"), args)
            if sp["compute0_us"] >= 0 and sp["compute1_us"] >= sp["compute0_us"]:
                tracer.span(f"stage{stage}.gpu.mb{int(sp['mb'])}", "compute", ns("compute0_us"), ns("compute1_us"), args)
            if sp["send1_us"] >= 0 and sp["send1_us"] >= sp["compute1_us"] >= 0:
                tracer.span(f"stage{stage}.hop-out", "send", ns("compute1_us"), ns("send1_us"), args)


class NativeReplay(_LoopCommon):
    """`cin` / `res`: the `ShmRing`s of the descriptor and result rings (this process consumes the
    first and is the only producer of the second)."""

    def __init__(self, device_index: int, max_mbs: int, cin, res):
        from ..utils import native
        self.L = _lib()
        rt = native.lib()
        self.cin, self.res = cin, res
        self.h = self.L.dlia_replay_create(device_index, max_mbs, C.cast(rt.dlia_ring_get, C.c_void_p),
                                           C.cast(rt.dlia_ring_put, C.c_void_p))
        if not self.h:
            raise NativeReplayError("cannot create the native serving loop")
        self._buf = C.create_string_buffer(cin.slot_bytes)
        self._len = C.c_uint32(0)

    def _check(self, rc: int, what: str) -> int:
        if rc == -1:
            raise NativeReplayError(f"{what}: {self.L.dlia_replay_error(self.h).decode(errors='replace')}")
        return rc

    def set_table(self, table, num_blocks: int):
        self._check(self.L.dlia_replay_set_table(self.h, table.data_ptr(), table.shape[0], table.shape[1], num_blocks), "set_table")

    def set_mb(self, mb: int, graph_exec: int, stream: int, tokens_ptr: int, rows: int):
        self._check(self.L.dlia_replay_set_mb(self.h, mb, C.c_void_p(graph_exec), C.c_void_p(stream), C.c_void_p(tokens_ptr), rows),
                    "set_mb")

    def push_external(self, step_id: int, event: Optional[int], host_ptr: int, n: int):
        self.L.dlia_replay_push_external(self.h, step_id, C.c_void_p(event) if event else None, C.c_void_p(host_ptr), n)

    def serve(self, idle_ms: int) -> Tuple[int, Optional[bytes]]:
        """(1, message for Python) | (0, None) after `idle_ms` without one | (-2, None) ring closed."""
        rc = self._check(self.L.dlia_replay_serve(self.h, self.cin._h, self.res._h, self._buf, self.cin.slot_bytes,
                                                  C.byref(self._len), idle_ms), "native serving loop")
        return rc, (self._buf.raw[:self._len.value] if rc == 1 else None)

    def flush(self, timeout_s: float = 30.0) -> bool:
        return self._check(self.L.dlia_replay_flush(self.h, self.res._h, int(timeout_s * 1000)), "flush") == 0

    def close(self):
        if self.h:
            self.L.dlia_replay_destroy(self.h)
            self.h = None


class NativeHeadLoop(NativeReplay):
    """The native loop of stage 0 of a multi-stage pipeline: replay descriptors from the scheduler's
    ring -> wait for the micro-batch's fed-back ids -> hipGraphLaunch -> descriptor on -> send of the
    hidden states -> receive of the step's ids from the last stage -> pinned host copy -> result ring.
    `next_ring`: stage 1's descriptor ring; `edge_fb` / `edge_out`: the token feedback hop (receiving
    end) and the hop to stage 1 (sending end), objects with `native_ends()`."""

    def __init__(self, device_index: int, max_mbs: int, cin, res, next_ring, edge_fb, edge_out):
        super().__init__(device_index, max_mbs, cin, res)
        self.next_ring = next_ring
        ih, _, irecv, _, iwait, ierr = edge_fb.native_ends()
        oh, osend, _, owait, _, oerr = edge_out.native_ends()
        out_stream = edge_out.native_send_stream() if hasattr(edge_out, "native_send_stream") else 0
        self.L.dlia_stage_set_edges(self.h, ih, irecv, iwait, ierr, oh, osend, owait, oerr, C.c_void_p(out_stream))
        self._keep = (edge_fb, edge_out)

    def set_mb(self, mb: int, graph_exec: int, stream: int, tokens, out, rows: int):
        """`tokens`: the graph's static token-id input (int32 [rows]); `out`: its hidden-state output."""
        super().set_mb(mb, graph_exec, stream, tokens.data_ptr(), rows)
        self._check(self.L.dlia_head_set_out(self.h, mb, C.c_void_p(out.data_ptr()), out.numel() * out.element_size()), "set_out")

    def serve(self, idle_ms: int) -> Tuple[int, Optional[bytes]]:
        rc = self._check(self.L.dlia_head_serve(self.h, self.cin._h, self.res._h, self.next_ring._h, self._buf, self.cin.slot_bytes,
                                                C.byref(self._len), idle_ms), "native stage-0 loop")
        return rc, (self._buf.raw[:self._len.value] if rc == 1 else None)


class NativeStageLoop(_LoopCommon):
    """The native loop of a stage behind stage 0: replay descriptors from the previous stage's ring
    -> receive -> hipGraphLaunch -> descriptor on -> send, without entering Python. `prev_ring` /
    `next_ring`: `ShmRing`s (`next_ring` None on the last stage); `edge_in` / `edge_out`: hop objects
    with `native_ends()` (`RcclEdge`, `LocalDeviceEdge`) — on the last stage `edge_out` is the token
    feedback hop."""

    def __init__(self, device_index: int, max_mbs: int, prev_ring, next_ring, edge_in, edge_out):
        from ..utils import native
        self.L = _lib()
        rt = native.lib()
        self.prev_ring, self.next_ring = prev_ring, next_ring
        self.h = self.L.dlia_replay_create(device_index, max_mbs, C.cast(rt.dlia_ring_get, C.c_void_p),
                                           C.cast(rt.dlia_ring_put, C.c_void_p))
        if not self.h:
            raise NativeReplayError("cannot create the native stage loop")
        ih, _, irecv, _, iwait, ierr = edge_in.native_ends()
        oh, osend, _, owait, _, oerr = edge_out.native_ends()
        out_stream = edge_out.native_send_stream() if hasattr(edge_out, "native_send_stream") else 0
        self.L.dlia_stage_set_edges(self.h, ih, irecv, iwait, ierr, oh, osend, owait, oerr, C.c_void_p(out_stream))
        self._keep = (edge_in, edge_out)
        self._buf = C.create_string_buffer(prev_ring.slot_bytes)
        self._len = C.c_uint32(0)

    def _check(self, rc: int, what: str) -> int:
        if rc == -1:
            raise NativeReplayError(f"{what}: {self.L.dlia_replay_error(self.h).decode(errors='replace')}")
        return rc

    def set_table(self, table, num_blocks: int):
        self._check(self.L.dlia_replay_set_table(self.h, table.data_ptr(), table.shape[0], table.shape[1], num_blocks), "set_table")

    def set_mb(self, mb: int, graph_exec: int, stream: int, inp, out, rows: int):
        """`inp` / `out`: the graph's static input and output tensors."""
        self._check(self.L.dlia_stage_set_mb(self.h, mb, C.c_void_p(graph_exec), C.c_void_p(stream), C.c_void_p(inp.data_ptr()),
                                             inp.numel() * inp.element_size(), C.c_void_p(out.data_ptr()),
                                             out.numel() * out.element_size(), rows), "set_mb")

    def clear_mb(self, mb: int):
        self.L.dlia_stage_set_mb(self.h, mb, None, None, None, 0, None, 0, 0)

    def serve(self, idle_ms: int) -> Tuple[int, Optional[bytes]]:
        """(1, message for Python) | (0, None) after `idle_ms` | (-2, None) ring closed."""
        rc = self._check(self.L.dlia_stage_serve(self.h, self.prev_ring._h, self.next_ring._h if self.next_ring is not None else None,
                                                 self._buf, self.prev_ring.slot_bytes, C.byref(self._len), idle_ms),
                         "native stage loop")
        return rc, (self._buf.raw[:self._len.value] if rc == 1 else None)

    def close(self):
        if self.h:
            self.L.dlia_replay_destroy(self.h)
            self.h = None
