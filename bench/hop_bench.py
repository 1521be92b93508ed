#!/usr/bin/env python3
"""Cost of one pipeline hop through the native RCCL edge, measured on ONE GPU with a one-rank
communicator exchanging with itself (the wire path minus the second device):

* host time per hop: what the stage loop pays to enqueue fence + transfer + completion event;
* device latency per hop: back-to-back dependent hops, per message size (decode rows x hidden).

    python bench/hop_bench.py [--iters 200]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_llm_inference_amd.parallel import rccl      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    edge = rccl.RcclEdge(rccl.unique_id(), 1, 0, 0, "cuda:0", name="self")
    out = {"rccl_version": rccl.version(), "rows": []}
    for rows, hidden in ((1, 4096), (64, 4096), (256, 4096), (1024, 4096), (1024, 8192), (1024, 1)):
        dt = torch.int32 if hidden == 1 else torch.bfloat16
        src = torch.ones(rows, hidden, device="cuda", dtype=dt)
        dst = torch.zeros_like(src)
        for _ in range(10):
            edge.sendrecv(src, dst).wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            edge.sendrecv(src, dst).wait()          # dependent chain: each hop waits for the previous
        host = (time.perf_counter() - t0) / a.iters
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.iters
        nbytes = src.numel() * src.element_size()
        out["rows"].append({"rows": rows, "hidden": hidden, "bytes": nbytes, "host_us_per_hop": round(host * 1e6, 2),
                            "device_us_per_hop": round(wall * 1e6, 2),
                            "GBps": round(nbytes / wall / 1e9, 2)})
    edge.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
