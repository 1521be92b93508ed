"""Step timeline tracing (Chrome trace / Perfetto JSON) and structured logging.

The reference's only timing is one `time.time()` pair around the whole request
(orchestration.py:82,201) and emoji `print` breadcrumbs. Here every stage can record spans
(enqueue / recv-wait / compute / send) and the orchestrator can dump them with `--trace`.
"""
from __future__ import annotations

import json
import logging
import os
import threading
import time
from typing import Dict, List, Optional


def get_logger(name: str = "dlia") -> logging.Logger:
    log = logging.getLogger(name)
    if not log.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("%(asctime)s %(levelname)s %(name)s | %(message)s", "%H:%M:%S"))
        log.addHandler(h)
        log.setLevel(os.environ.get("DLIA_LOG", "INFO").upper())
        log.propagate = False
    return log


class Tracer:
    """Collects complete ("X") events; thread-safe; bounded."""

    def __init__(self, max_events: int = 200_000):
        self.events: List[Dict] = []
        self.max_events = max_events
        self._lock = threading.Lock()
        self._pid = os.getpid()

    def span(self, track: str, name: str, t0_ns: int, t1_ns: int, args: Optional[Dict] = None):
        if len(self.events) >= self.max_events:
            return
        ev = {"name": name, "cat": track, "ph": "X", "pid": self._pid, "tid": track,
              "ts": t0_ns / 1e3, "dur": max(0.0, (t1_ns - t0_ns) / 1e3), "args": args or {}}
        with self._lock:
            self.events.append(ev)

    def instant(self, track: str, name: str, args: Optional[Dict] = None):
        ev = {"name": name, "cat": track, "ph": "i", "s": "t", "pid": self._pid, "tid": track,
              "ts": time.perf_counter_ns() / 1e3, "args": args or {}}
        with self._lock:
            if len(self.events) < self.max_events:
                self.events.append(ev)

    def dump(self, path: str):
        with self._lock:
            data = {"traceEvents": list(self.events), "displayTimeUnit": "ms"}
        with open(path, "w") as f:
            json.dump(data, f)
        return path
