"""ctypes front of the native RCCL hop (`ops/csrc/comm/rccl_edge.cpp` -> `ops/_comm.so`).

`RcclEdge` is one directed pipeline hop with its own communicator and HIP stream: hidden states go
GPU to GPU over the xGMI link between two neighbouring stages, the sampled token ids go back from
the last stage to stage 0. It replaces the reference's JSON-over-HTTP relay
(orchestration.py:114-137; Worker1.py:208-245). Nothing here blocks the host: `send` / `recv`
enqueue the transfer on the edge's stream behind an event of the producing (or previously reading)
stream and return a handle; `handle.wait()` makes the current stream wait for the transfer;
`handle.sync(timeout)` is the only host-side wait and it has a deadline.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch

from .transport import Edge, Handle

_lock = threading.Lock()
_lib = None


class RcclError(RuntimeError):
    pass


def limit_p2p_channels():
    """RCCL's point-to-point kernels take one workgroup per channel — 32 of them for the 16.8 MB of
    a 2048-row hidden state with the defaults (measured on MI355X, `bench/debug/hop_grid.sh`), 132
    registers per lane each. A receive is posted as soon as its descriptor arrives and then spins
    until the previous stage has computed, for milliseconds: every CU it sits on cannot take a GEMM
    workgroup meanwhile (those need the whole register file), 12 % of the chip. Four channels move
    the same message at ~80 GB/s on one device (one: 22, two: 43), more than an xGMI link carries,
    and keep the spinning to 4 CUs. Read by RCCL when it first initialises; an explicit setting in
    the environment wins."""
    os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", "4")
    os.environ.setdefault("NCCL_MIN_P2P_NCHANNELS", "1")


def lib():
    """Loads `_comm.so` and points it at the RCCL library of this PyTorch build."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        limit_p2p_channels()
        from ..ops import build as _b
        path = os.path.join(os.path.dirname(_b.__file__), "_comm.so")
        if not os.path.exists(path):
            _b.build_comm(verbose=False)
        L = C.CDLL(path)
        vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
        sig = {
            "dlia_comm_last_error": (C.c_char_p, []),
            "dlia_comm_init": (ci, [C.c_char_p]),
            "dlia_comm_version": (ci, []),
            "dlia_comm_unique_id": (ci, [vp]),
            "dlia_edge_open": (ci, [vp, ci, ci, ci, ci, C.POINTER(vp)]),
            "dlia_edge_send": (ci, [vp, vp, sz, vp]),
            "dlia_edge_recv": (ci, [vp, vp, sz, vp, vp]),
            "dlia_edge_sendrecv": (ci, [vp, vp, vp, sz, vp]),
            "dlia_edge_send_on": (ci, [vp, vp, sz, vp]),
            "dlia_edge_recv_on": (ci, [vp, vp, sz, vp]),
            "dlia_edge_sendrecv_on": (ci, [vp, vp, vp, sz, vp]),
            "dlia_edge_wait": (ci, [vp, ci, vp]),
            "dlia_edge_query": (ci, [vp, ci]),
            "dlia_edge_sync": (ci, [vp, ci, ci]),
            "dlia_edge_stream": (vp, [vp]),
            "dlia_edge_stats": (None, [vp, C.POINTER(C.c_uint64)]),
            "dlia_edge_abort": (ci, [vp]),
            "dlia_edge_close": (ci, [vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if L.dlia_comm_init(rccl.encode()) != 0:
            raise RcclError(f"cannot bind RCCL at {rccl}: {L.dlia_comm_last_error().decode()}")
        _lib = L
        return L


def _check(rc: int, what: str) -> int:
    if rc < 0:
        raise RcclError(f"{what}: {lib().dlia_comm_last_error().decode(errors='replace')}")
    return rc


def version() -> int:
    return int(lib().dlia_comm_version())


def unique_id() -> bytes:
    """128 bytes identifying one communicator; created by one rank, handed to all of its ranks."""
    buf = C.create_string_buffer(128)
    _check(lib().dlia_comm_unique_id(buf), "ncclGetUniqueId")
    return buf.raw


def _stream_ptr(device) -> C.c_void_p:
    # the default stream's handle is 0, which ctypes would otherwise pass as "None"
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class RcclHandle(Handle):
    def __init__(self, edge: "RcclEdge", token: int, keep=None):
        self.edge, self.token, self.keep = edge, token, keep
        self._waited = False

    def wait(self):
        """The current stream waits for the transfer (the host does not)."""
        e = self.edge
        _check(e.L.dlia_edge_wait(e.h, self.token, _stream_ptr(e.device)), "edge wait")
        self._waited = True

    def done(self) -> bool:
        return _check(self.edge.L.dlia_edge_query(self.edge.h, self.token), "edge query") == 1

    def sync(self, timeout_s: float = 30.0):
        rc = _check(self.edge.L.dlia_edge_sync(self.edge.h, self.token, int(timeout_s * 1000)), "edge sync")
        if rc == 1:
            raise TimeoutError(f"RCCL hop {self.edge.name}: no completion within {timeout_s:.1f}s")


class RcclEdge(Edge):
    """One directed hop. `rank` / `peer` are ranks inside this edge's own communicator (0 and 1 for
    a two-process hop; 0 and 0 for the single-process self hop used to exercise the wire path on
    one GPU). Opening is collective over the communicator's ranks."""

    def __init__(self, uid: bytes, nranks: int, rank: int, peer: int, device, name: str = "edge"):
        self.L = lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RcclError("RCCL edges connect GPUs; use gloo / loopback on CPU")
        self.name = name
        self.nranks, self.rank, self.peer = nranks, rank, peer
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        h = C.c_void_p()
        _check(self.L.dlia_edge_open(uid, nranks, rank, peer, idx, C.byref(h)), f"open RCCL hop {name}")
        self.h = h
        self.closed = False

    # ---- Edge interface ---------------------------------------------------------------------
    def send(self, t: torch.Tensor) -> RcclHandle:
        t = t.contiguous()
        tok = _check(self.L.dlia_edge_send(self.h, t.data_ptr(), t.numel() * t.element_size(),
                                           _stream_ptr(self.device)), f"send on {self.name}")
        return RcclHandle(self, tok, keep=t)

    def recv(self, buf: torch.Tensor, after_event: Optional[torch.cuda.Event] = None) -> RcclHandle:
        """Receive into `buf`. The transfer is ordered after the work already enqueued on the
        current stream (the previous reader of `buf`) and after `after_event` when given."""
        assert buf.is_contiguous()
        ev = after_event.cuda_event if after_event is not None else None
        tok = _check(self.L.dlia_edge_recv(self.h, buf.data_ptr(), buf.numel() * buf.element_size(),
                                           _stream_ptr(self.device), ev), f"recv on {self.name}")
        return RcclHandle(self, tok, keep=buf)

    def sendrecv(self, src: torch.Tensor, dst: torch.Tensor) -> RcclHandle:
        assert src.is_contiguous() and dst.is_contiguous() and src.numel() * src.element_size() == dst.numel() * dst.element_size()
        tok = _check(self.L.dlia_edge_sendrecv(self.h, src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size(),
                                               _stream_ptr(self.device)), f"sendrecv on {self.name}")
        return RcclHandle(self, tok, keep=(src, dst))

    # ---- on the caller's stream (capturable) --------------------------------------------------
    def send_on_current_stream(self, t: torch.Tensor):
        _check(self.L.dlia_edge_send_on(self.h, t.data_ptr(), t.numel() * t.element_size(), _stream_ptr(self.device)),
               f"send on {self.name}")

    def recv_on_current_stream(self, buf: torch.Tensor):
        _check(self.L.dlia_edge_recv_on(self.h, buf.data_ptr(), buf.numel() * buf.element_size(), _stream_ptr(self.device)),
               f"recv on {self.name}")

    def sendrecv_on_current_stream(self, src: torch.Tensor, dst: torch.Tensor):
        _check(self.L.dlia_edge_sendrecv_on(self.h, src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size(),
                                            _stream_ptr(self.device)), f"sendrecv on {self.name}")

    def native_ends(self):
        """(handle, send, recv, wait for a send, wait for a receive, last error text) for the native stage loop."""
        f = lambda name: C.cast(getattr(self.L, name), C.c_void_p)
        return self.h, f("dlia_edge_send"), f("dlia_edge_recv"), f("dlia_edge_wait"), f("dlia_edge_wait"), f("dlia_comm_last_error")

    def native_send_stream(self) -> int:
        """Handle of the HIP stream this hop's transfers run on."""
        return int(self.L.dlia_edge_stream(self.h) or 0)

    # ---- lifecycle ---------------------------------------------------------------------------
    def stats(self):
        out = (C.c_uint64 * 4)()
        self.L.dlia_edge_stats(self.h, out)
        return {"bytes_sent": int(out[0]), "bytes_recv": int(out[1]), "ops": int(out[2]), "ring_waits": int(out[3])}

    def abort(self):
        """Release transfers stuck on a peer that will never answer; only `close` is valid after."""
        if not self.closed:
            _check(self.L.dlia_edge_abort(self.h), f"abort {self.name}")

    def close(self):
        if not self.closed:
            self.closed = True
            self.L.dlia_edge_close(self.h)
