"""ctypes front of the in-process device hop (`ops/csrc/comm/local_edge.cpp` -> `ops/_comm.so`).

`LocalDeviceEdge` joins two stages that live in one process on one GPU (the stage threads of a
`LocalPipeline`): a send copies into a device staging ring behind the producer's work, a receive
copies out behind the buffer's previous reader, the consumer's stream waits for the receive's
event. It has the C ABI of the RCCL hop, so the native stage loop (`runtime/replay.py`) drives
either; that is how the loop's recv -> graph -> send path runs on a one-GPU box. The reference's
hop is JSON over HTTP through the orchestrator (orchestration.py:114-137; Worker1.py:208-245).
"""
from __future__ import annotations

import ctypes as C

import torch

from .transport import Edge, Handle

_sigs_done = False


def _lib():
    global _sigs_done
    from . import rccl
    L = rccl.lib()
    if not _sigs_done:
        vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
        sig = {
            "dlia_ledge_last_error": (C.c_char_p, []),
            "dlia_ledge_open": (ci, [ci, sz, C.POINTER(vp)]),
            "dlia_ledge_send": (ci, [vp, vp, sz, vp]),
            "dlia_ledge_recv": (ci, [vp, vp, sz, vp, vp]),
            "dlia_ledge_wait": (ci, [vp, ci, ci, vp]),
            "dlia_ledge_query": (ci, [vp, ci, ci]),
            "dlia_ledge_set_timeout": (None, [vp, ci]),
            "dlia_ledge_pending": (ci, [vp]),
            "dlia_ledge_send_stream": (vp, [vp]),
            "dlia_ledge_abort": (ci, [vp]),
            "dlia_ledge_close": (ci, [vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _sigs_done = True
    return L


class LocalEdgeError(RuntimeError):
    pass


class _LocalHandle(Handle):
    def __init__(self, edge: "LocalDeviceEdge", token: int, recv_side: int, keep=None):
        self.edge, self.token, self.side, self.keep = edge, token, recv_side, keep

    def wait(self):
        e = self.edge
        if e.L.dlia_ledge_wait(e.h, self.token, self.side, C.c_void_p(torch.cuda.current_stream(e.device).cuda_stream)) != 0:
            raise LocalEdgeError(e.L.dlia_ledge_last_error().decode(errors="replace"))

    def done(self) -> bool:
        rc = self.edge.L.dlia_ledge_query(self.edge.h, self.token, self.side)
        if rc < 0:
            raise LocalEdgeError(self.edge.L.dlia_ledge_last_error().decode(errors="replace"))
        return rc == 1


class LocalDeviceEdge(Edge):
    """Both ends of one hop (the sender's thread calls `send`, the receiver's `recv`)."""

    def __init__(self, device, max_bytes: int, timeout_s: float = 60.0, name: str = "local"):
        self.L = _lib()
        self.device = torch.device(device)
        self.name = name
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        h = C.c_void_p()
        if self.L.dlia_ledge_open(idx, max(int(max_bytes), 256), C.byref(h)) != 0:
            raise LocalEdgeError(f"open local hop {name}: {self.L.dlia_ledge_last_error().decode(errors='replace')}")
        self.h = h
        self.L.dlia_ledge_set_timeout(self.h, int(timeout_s * 1000))
        self.closed = False

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def send(self, t: torch.Tensor) -> _LocalHandle:
        t = t.contiguous()
        tok = self.L.dlia_ledge_send(self.h, t.data_ptr(), t.numel() * t.element_size(), self._stream())
        if tok < 0:
            raise LocalEdgeError(f"send on {self.name}: {self.L.dlia_ledge_last_error().decode(errors='replace')}")
        return _LocalHandle(self, tok, 0, keep=t)

    def recv(self, buf: torch.Tensor) -> _LocalHandle:
        """Blocks the calling thread (not the device) until the matching send has been enqueued."""
        assert buf.is_contiguous()
        tok = self.L.dlia_ledge_recv(self.h, buf.data_ptr(), buf.numel() * buf.element_size(), self._stream(), None)
        if tok < 0:
            raise LocalEdgeError(f"recv on {self.name}: {self.L.dlia_ledge_last_error().decode(errors='replace')}")
        return _LocalHandle(self, tok, 1, keep=buf)

    def pending(self) -> int:
        """Messages already sent (enqueued) and not yet received: `recv` does not block while this is > 0."""
        return int(self.L.dlia_ledge_pending(self.h))

    def native_ends(self):
        """(handle, send, recv, wait for a send, wait for a receive, last error text) for the native stage loop."""
        f = lambda name: C.cast(getattr(self.L, name), C.c_void_p)
        return self.h, f("dlia_ledge_send"), f("dlia_ledge_recv"), f("dlia_ledge_wait_send"), f("dlia_ledge_wait_recv"), f("dlia_ledge_last_error")

    def native_send_stream(self) -> int:
        """Handle of the HIP stream this hop's sends run on."""
        return int(self.L.dlia_ledge_send_stream(self.h) or 0)

    def abort(self):
        if not self.closed:
            self.L.dlia_ledge_abort(self.h)

    def close(self):                      # both ends hold the object: the first close wins
        if not self.closed:
            self.closed = True
            self.L.dlia_ledge_close(self.h)
