"""Shared-memory control plane (native, `ops/csrc/runtime`): SPSC message rings between
neighbouring stages and the per-stage status board read by `/workers`."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

from ..utils import native

STATE_NAMES = {0: "unregistered", 1: "starting", 2: "ready", 3: "error", 4: "stopped"}
ST_STARTING, ST_READY, ST_ERROR, ST_STOPPED = 1, 2, 3, 4


class RingClosed(Exception):
    pass


class ShmRing:
    """One direction of one pipeline edge. Exactly one producer process and one consumer."""

    def __init__(self, name: str, create: bool, slots: int = 256, slot_bytes: int = 65536, timeout_s: float = 300.0):
        self._L = native.lib()
        self.name = name if name.startswith("/") else "/" + name
        self._h = self._L.dlia_ring_open(self.name.encode(), 1 if create else 0, slots, slot_bytes, int(timeout_s * 1000))
        if not self._h:
            raise OSError(f"cannot {'create' if create else 'attach to'} shared-memory ring {self.name}")
        self._owner = create
        self.slot_bytes = int(self._L.dlia_ring_slot_bytes(self._h))
        self._buf = C.create_string_buffer(self.slot_bytes)

    def put(self, msg: bytes, timeout_s: Optional[float] = None):
        rc = self._L.dlia_ring_put(self._h, msg, len(msg), -1 if timeout_s is None else int(timeout_s * 1000))
        if rc == 1:
            raise TimeoutError(f"ring {self.name} full")
        if rc == 2:
            raise ValueError(f"message of {len(msg)} bytes exceeds slot size {self.slot_bytes}")
        if rc == 3:
            raise RingClosed(self.name)

    def get(self, timeout_s: Optional[float] = None) -> Optional[bytes]:
        """Next message, or None on timeout."""
        n = self._L.dlia_ring_get(self._h, self._buf, self.slot_bytes, -1 if timeout_s is None else int(timeout_s * 1000))
        if n == -1:
            return None
        if n == -3:
            raise RingClosed(self.name)
        if n < 0:
            raise RuntimeError(f"ring {self.name} get failed ({n})")
        return self._buf.raw[:n]

    def pending(self) -> int:
        return int(self._L.dlia_ring_pending(self._h))

    def mark_closed(self):
        if self._h:
            self._L.dlia_ring_mark_closed(self._h)

    def close(self):
        if self._h:
            self._L.dlia_ring_close(self._h, 1 if self._owner else 0)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StatusBoard:
    def __init__(self, name: str, create: bool, num_stages: int = 0, timeout_s: float = 300.0):
        self._L = native.lib()
        self.name = name if name.startswith("/") else "/" + name
        self._h = self._L.dlia_board_open(self.name.encode(), 1 if create else 0, num_stages, int(timeout_s * 1000))
        if not self._h:
            raise OSError(f"cannot open status board {self.name}")
        self._owner = create
        self.num_stages = int(self._L.dlia_board_num_stages(self._h))

    def register(self, stage: int, layer_start: int, layer_end: int, pid: Optional[int] = None):
        self._L.dlia_board_register(self._h, stage, pid or os.getpid(), layer_start, layer_end)

    def set_state(self, stage: int, state: int, error: str = ""):
        self._L.dlia_board_set_state(self._h, stage, state, error.encode()[:63] if error else None)

    def beat(self, stage: int, steps: int = 0, tokens: int = 0, busy_ns: int = 0):
        self._L.dlia_board_beat(self._h, stage, steps, tokens, busy_ns)

    def set_kv_blocks(self, stage: int, blocks: int):
        self._L.dlia_board_set_kv(self._h, stage, int(blocks))

    def agree_kv_blocks(self, stage: int, mine: int, timeout_s: float = 600.0) -> int:
        """Publishes this stage's KV block budget and returns the minimum over all stages once every
        stage has published its own: block ids are global to the pipeline, so every cache holds
        exactly that many."""
        import time
        self.set_kv_blocks(stage, mine)
        t0 = time.time()
        while True:
            vals = [self.read(s) for s in range(self.num_stages)]
            if any(v["state"] == "error" for v in vals):
                raise RuntimeError("a stage failed during bring-up: " + "; ".join(v["error"] for v in vals if v["error"]))
            if vals and all(v["kv_blocks"] > 0 for v in vals):
                return min(v["kv_blocks"] for v in vals)
            if time.time() - t0 > timeout_s:
                raise TimeoutError("stages did not agree on a KV block count")
            time.sleep(0.01)

    def read(self, stage: int) -> Dict:
        out = (C.c_int64 * 12)()
        err = C.create_string_buffer(64)
        self._L.dlia_board_read(self._h, stage, out, err, 64)
        return {"state": STATE_NAMES.get(int(out[0]), "?"), "pid": int(out[1]), "age_ms": int(out[2]),
                "steps": int(out[3]), "tokens": int(out[4]), "busy_ns": int(out[5]),
                "layers": f"{int(out[6])}-{int(out[7])}", "kv_blocks": int(out[8]),
                "error": err.value.decode(errors="replace")}

    def close(self):
        if self._h:
            self._L.dlia_board_close(self._h, 1 if self._owner else 0)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
