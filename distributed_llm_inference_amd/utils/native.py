"""ctypes access to the native host runtime (`ops/_rt.so`: shared-memory ring, status board,
KV block allocator). Built on first use when a compiler is present."""
from __future__ import annotations

import ctypes as C
import os
import threading

_lock = threading.Lock()
_lib = None


def lib():
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        from ..ops import build as _b
        path = os.path.join(os.path.dirname(_b.__file__), "_rt.so")
        if not os.path.exists(path):
            _b.build_runtime(verbose=False)
        L = C.CDLL(path)
        vp, i32, u32, i64, cp = C.c_void_p, C.c_int32, C.c_uint32, C.c_int64, C.c_char_p
        sig = {
            "dlia_ring_open": (vp, [cp, C.c_int, u32, u32, C.c_int]),
            "dlia_ring_slot_bytes": (u32, [vp]),
            "dlia_ring_put": (C.c_int, [vp, vp, u32, C.c_int]),
            "dlia_ring_get": (C.c_int, [vp, vp, u32, C.c_int]),
            "dlia_ring_pending": (C.c_int, [vp]),
            "dlia_ring_mark_closed": (None, [vp]),
            "dlia_ring_close": (None, [vp, C.c_int]),
            "dlia_board_open": (vp, [cp, C.c_int, u32, C.c_int]),
            "dlia_board_num_stages": (C.c_int, [vp]),
            "dlia_board_register": (None, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
            "dlia_board_set_state": (None, [vp, C.c_int, C.c_int, cp]),
            "dlia_board_beat": (None, [vp, C.c_int, i64, i64, i64]),
            "dlia_board_set_kv": (None, [vp, C.c_int, i64]),
            "dlia_board_read": (None, [vp, C.c_int, C.POINTER(i64), cp, C.c_int]),
            "dlia_board_close": (None, [vp, C.c_int]),
            "dlia_alloc_create": (vp, [i32]),
            "dlia_alloc_destroy": (None, [vp]),
            "dlia_alloc_take": (C.c_int, [vp, i64, i32, C.POINTER(i32)]),
            "dlia_alloc_release": (C.c_int, [vp, i64]),
            "dlia_alloc_free_count": (i32, [vp]),
            "dlia_alloc_owned_count": (i32, [vp, i64]),
            "dlia_alloc_peak_used": (i64, [vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
        return L
