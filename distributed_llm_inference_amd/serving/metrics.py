"""Prometheus text exposition of engine and pipeline-stage counters (`GET /metrics`).

The reference's only metrics are three strings in the `/generate` response body
(orchestration.py:201-218)."""
from __future__ import annotations

import time
from typing import Dict, Optional

_last_busy: Dict[tuple, tuple] = {}       # (board id, stage) -> (busy ns, monotonic ns) at the previous scrape


def render(engine_snapshot: Dict, board=None, num_stages: int = 0) -> str:
    lines = []

    def metric(name, value, help_, typ="gauge", labels: Optional[Dict] = None):
        if value is None:
            return
        if not any(l.startswith(f"# HELP dlia_{name} ") for l in lines):
            lines.append(f"# HELP dlia_{name} {help_}")
            lines.append(f"# TYPE dlia_{name} {typ}")
        lab = ""
        if labels:
            lab = "{" + ",".join(f'{k}="{v}"' for k, v in labels.items()) + "}"
        lines.append(f"dlia_{name}{lab} {value}")

    def histogram(name, h, help_):
        if not h:
            return
        lines.append(f"# HELP dlia_{name} {help_}")
        lines.append(f"# TYPE dlia_{name} histogram")
        for bound, cum in h["buckets"]:
            le = "+Inf" if bound == float("inf") else f"{bound:g}"
            lines.append(f'dlia_{name}_bucket{{le="{le}"}} {cum}')
        lines.append(f"dlia_{name}_sum {h['sum']:.6f}")
        lines.append(f"dlia_{name}_count {h['count']}")

    s = engine_snapshot
    metric("requests_total", s.get("requests"), "requests accepted", "counter")
    metric("requests_completed_total", s.get("completed"), "requests completed", "counter")
    metric("requests_failed_total", s.get("failed"), "requests failed", "counter")
    metric("requests_cancelled_total", s.get("cancelled"), "requests abandoned by their client (disconnect or timeout)", "counter")
    metric("requests_rejected_total", s.get("rejected"), "requests refused because the queue was full", "counter")
    metric("tokens_generated_total", s.get("tokens_out"), "output tokens", "counter")
    metric("prompt_tokens_total", s.get("tokens_in"), "prompt tokens", "counter")
    metric("steps_total", s.get("steps"), "pipeline steps issued", "counter")
    metric("prefill_steps_total", s.get("prefill_steps"), "prefill steps issued", "counter")
    metric("decode_steps_total", s.get("decode_steps"), "decode steps issued", "counter")
    metric("recompositions_total", s.get("recompositions"), "micro-batch recompositions", "counter")
    metric("requests_waiting", s.get("waiting"), "requests queued")
    metric("requests_running", s.get("running"), "sequences decoding")
    metric("preemptions_total", s.get("preemptions"), "sequences preempted for KV blocks (recomputed later)", "counter")
    metric("rows_moved_total", s.get("rows_moved"), "running sequences moved to lower rows so that decode steps could shrink", "counter")
    metric("kv_blocks_grown_total", s.get("kv_blocks_grown"), "KV blocks handed out on demand during decode", "counter")
    metric("kv_blocks_used", s.get("kv_blocks_used"), "KV blocks in use")
    metric("kv_blocks_total", s.get("kv_blocks_total"), "KV blocks per stage")
    metric("token_latency_ms", s.get("tpot_ms_p50"), "per-token decode latency", labels={"quantile": "0.5"})
    metric("token_latency_ms", s.get("tpot_ms_p99"), "per-token decode latency", labels={"quantile": "0.99"})
    histogram("time_to_first_token_seconds", s.get("ttft_hist"), "arrival of a request to its first token")
    histogram("inter_token_latency_seconds", s.get("itl_hist"), "gap between consecutive tokens of a request")
    if board is not None:
        now = time.monotonic_ns()
        for st in range(num_stages):
            r = board.read(st)
            lab = {"stage": st, "layers": r["layers"]}
            metric("stage_up", 1 if r["state"] == "ready" else 0, "stage is ready", labels=lab)
            metric("stage_steps_total", r["steps"], "steps executed by the stage", "counter", lab)
            metric("stage_tokens_total", r["tokens"], "tokens processed by the stage", "counter", lab)
            metric("stage_busy_seconds_total", r["busy_ns"] / 1e9,
                   "time the stage's GPU spent inside step kernels (HIP events around every step; union over its streams)", "counter", lab)
            prev = _last_busy.get((id(board), st))
            _last_busy[(id(board), st)] = (r["busy_ns"], now)
            if prev is not None and now > prev[1]:
                metric("stage_gpu_busy_fraction", round(min(1.0, max(0.0, (r["busy_ns"] - prev[0]) / (now - prev[1]))), 4),
                       "share of the time since the previous scrape the stage's GPU was inside step kernels", labels=lab)
            metric("stage_heartbeat_age_ms", r["age_ms"], "age of the last heartbeat", labels=lab)
    return "\n".join(lines) + "\n"
