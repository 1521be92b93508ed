"""One decode-regime GEMM shape, a few launches: the target of counter (PMC) passes.

    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum ... -- python3 bench/one_gemm.py --m 256 --n 28672 --k 4096 --epi 4
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_llm_inference_amd import ops
from distributed_llm_inference_amd.ops import packing


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=256)
    ap.add_argument("--n", type=int, default=28672)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--epi", type=int, default=4)
    ap.add_argument("--splitk", type=int, default=0)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--flush", action="store_true", help="evict L2 / Infinity Cache between launches (weights cold, as in a real step)")
    a = ap.parse_args()
    K = ops.load()
    x = torch.randn(a.m, a.k, device="cuda").to(torch.bfloat16)
    wp = packing.pack_weight((torch.randn(a.n, a.k, device="cuda") * 0.02).to(torch.bfloat16))
    n_out = a.n // 2 if a.epi == 4 else a.n
    out = torch.empty(a.m, n_out, device="cuda", dtype=torch.bfloat16)
    ws = torch.empty(16 * a.m * a.n, device="cuda", dtype=torch.float32) if a.m <= 256 else None   # split-K partials: decode regime only
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda") if a.flush else None
    evs = []
    for it in range(a.iters + 2):
        if flush is not None:
            flush.fill_(it & 1)
            x.add_(0)                      # x itself is warm in a real step: the previous kernel wrote it
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.gemm(out, x, wp, None, None, ws, a.n, a.epi, a.splitk, 0)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs[2:])
    print(f"M={a.m} N={a.n} K={a.k}: median {ts[len(ts) // 2]:.1f} us  min {ts[0]:.1f} us")


if __name__ == "__main__":
    main()
