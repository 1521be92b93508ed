"""Checks of a multi-GPU bring-up that can be read off one run: what devices and peer paths this
process sees, which RCCL it bound, and what one message costs on each pipeline hop.

`probe_hops` times every hop of an open pipeline (the chain stage i -> i+1, then the token feedback
hop) with the two message sizes the serving loop sends: a micro-batch's hidden states and its token
ids. `bench.py` stores the result in its JSON line, `bench.py --preflight` prints it without loading
a model. (The reference's only view of its hops is a 30 s timeout per POST: orchestration.py:118,131.)
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional

import torch


def device_report(peer_matrix: bool = True) -> Dict:
    """Devices visible to this process, their memory, the peer-access matrix and the RCCL version.
    Creates no device context beyond what the property queries need."""
    n = torch.cuda.device_count()
    rep: Dict = {"device_count": n, "devices": [], "rccl_version": None, "peer_access": None,
                 "env": {k: os.environ[k] for k in sorted(os.environ)
                         if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_VISIBLE", "ROCR_VISIBLE", "GPU_MAX_HW", "DLIA_"))}}
    for i in range(n):
        p = torch.cuda.get_device_properties(i)
        rep["devices"].append({"index": i, "name": p.name, "gcn_arch": getattr(p, "gcnArchName", ""),
                               "memory_gib": round(p.total_memory / 2 ** 30, 1), "cus": p.multi_processor_count})
    if peer_matrix and n > 1:
        rep["peer_access"] = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(n)] for i in range(n)]
    if n > 0:
        try:
            from . import rccl
            rep["rccl_version"] = rccl.version()
        except Exception as e:                     # noqa: BLE001 - reported, not fatal: the fallback transports do not need it
            rep["rccl_version"] = f"unavailable: {type(e).__name__}: {e}"
    return rep


def _finish(h, timeout_s: float, device):
    """Host-side completion of a transfer handle of any edge kind."""
    if hasattr(h, "sync"):
        h.sync(timeout_s)
        return
    h.wait()
    if device is not None:
        torch.cuda.current_stream(device).synchronize()


def _barrier(device):
    import torch.distributed as dist
    if dist.get_backend() == "nccl" and device is not None:
        dist.barrier(device_ids=[torch.device(device).index or 0])
    else:
        dist.barrier()


def probe_hops(links, device, big_bytes: int, small_bytes: int = 8192, reps: int = 3,
               timeout_s: float = 60.0) -> List[Dict]:
    """Raw one-way timings of every hop of an open multi-process pipeline, in the direction the
    serving loop uses it (nothing is ever sent the other way): for each hop — the chain stage
    i -> i+1, then the token feedback hop — and each message size (hidden states of a micro-batch =
    `big_bytes`, and `small_bytes`; the feedback hop only the small one), `reps` + 1 times: the
    receiver posts its receive, ALL stages meet at a barrier of the default group, the sender stamps
    the host clock and sends, the receiver stamps it when the bytes have landed. The ranks of one
    node share CLOCK_MONOTONIC, so arrival minus departure is the hop's latency with a waiting
    receiver — the situation of a stage whose predecessor finishes a micro-batch. Every stage must
    call this at the same point of its bring-up (the barriers are collective). Returns this
    stage's stamps; `combine_hops` pairs the two ends."""
    dev = torch.device(device) if device is not None else None
    me, n = links.stage, links.num_stages
    hops = [(s, s + 1, f"{s}->{s + 1}") for s in range(n - 1)] + ([(n - 1, 0, f"{n - 1}->0 (tokens)")] if n > 1 else [])
    out = []
    for src, dst, name in hops:
        edge = None
        if me == src:
            edge = links.fb_data if "tokens" in name else links.next_data
        elif me == dst:
            edge = links.fb_data if "tokens" in name else links.prev_data
        for nbytes in ([small_bytes] if "tokens" in name else [small_bytes, big_bytes]):
            stamps = []
            buf = None
            if edge is not None:
                buf = torch.zeros(max(4, nbytes) // 2, dtype=torch.float16, device=dev if dev is not None else "cpu")
                if dev is not None:
                    torch.cuda.current_stream(dev).synchronize()
            for _ in range(reps + 1):
                h = edge.recv(buf) if me == dst else None
                _barrier(dev)
                if me == src:
                    stamps.append(time.perf_counter())
                    _finish(edge.send(buf), timeout_s, dev)
                elif me == dst:
                    _finish(h, timeout_s, dev)
                    stamps.append(time.perf_counter())
            if edge is not None:
                out.append({"hop": name, "bytes": nbytes, "role": "src" if me == src else "dst", "stamps": stamps})
    return out


def combine_hops(per_rank: List[List[Dict]]) -> List[Dict]:
    """Pairs the senders' and receivers' stamps of `probe_hops` (one list per rank): per hop and
    message size the fastest and the median one-way time after the first (warm-up) message."""
    ends: Dict[tuple, Dict[str, list]] = {}
    for rows in per_rank:
        for r in rows or []:
            ends.setdefault((r["hop"], r["bytes"]), {})[r["role"]] = r["stamps"]
    out = []
    for (hop, nbytes), e in ends.items():
        if "src" not in e or "dst" not in e:
            continue
        d = sorted((b - a) * 1e6 for a, b in list(zip(e["src"], e["dst"]))[1:])
        if not d:
            continue
        out.append({"hop": hop, "bytes": nbytes, "us_min": round(d[0], 1), "us_p50": round(d[len(d) // 2], 1),
                    "GBps": round(nbytes / max(d[0], 1e-3) / 1e3, 2)})
    return out
