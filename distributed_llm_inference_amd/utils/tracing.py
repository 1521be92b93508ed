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
        self._gpu_base = None
        self._gpu_spans: List[tuple] = []

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

    # ---- device timeline: pairs of HIP events, resolved to host time at dump ------------------
    def gpu_begin(self):
        """Records a timing event on the current stream; returns a token for `gpu_end`."""
        import torch
        if self._gpu_base is None:
            base = torch.cuda.Event(enable_timing=True)
            base.record()
            base.synchronize()
            self._gpu_base = (base, time.perf_counter_ns())
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def gpu_end(self, token, track: str, name: str, args: Optional[Dict] = None):
        import torch
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        with self._lock:
            if len(self._gpu_spans) < self.max_events:
                self._gpu_spans.append((token, ev, track, name, args or {}))

    def _resolve_gpu(self):
        if not self._gpu_spans:
            return
        base, host_ns = self._gpu_base
        with self._lock:
            spans, self._gpu_spans = self._gpu_spans, []
        for e0, e1, track, name, args in spans:
            e1.synchronize()
            t0 = host_ns / 1e3 + base.elapsed_time(e0) * 1e3
            self.events.append({"name": name, "cat": track, "ph": "X", "pid": self._pid, "tid": track,
                                "ts": t0, "dur": max(0.0, e0.elapsed_time(e1) * 1e3), "args": args})

    def dump(self, path: str, merge_glob: Optional[str] = None):
        """Writes the Chrome-trace JSON. `merge_glob`: other stages' dumps to fold in (and delete)."""
        self._resolve_gpu()
        with self._lock:
            events = list(self.events)
        if merge_glob:
            import glob
            for other in sorted(glob.glob(merge_glob)):
                if os.path.abspath(other) == os.path.abspath(path):
                    continue
                try:
                    with open(other) as f:
                        events.extend(json.load(f).get("traceEvents", []))
                    os.unlink(other)
                except (OSError, ValueError):
                    pass
        with open(path, "w") as f:
            json.dump({"traceEvents": events, "displayTimeUnit": "ms"}, f)
        return path


def make_tracer(trace_path: Optional[str]) -> Optional[Tracer]:
    return Tracer() if trace_path else None


def stage_trace_path(trace_path: str, stage: int) -> str:
    return trace_path if stage == 0 else f"{trace_path}.stage{stage}"
