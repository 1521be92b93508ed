"""Where the time of a K-chunk goes in the loader-wave GEMM (gemm_rows.hip): runs the diagnostic
build of the kernel (in-kernel s_memtime stamps) on one shape and prints per-wave medians.

    DLIA_GEMM_ROWS_DIAG=1 python bench/gemm_diag.py --m 4096 --n 28672 --k 4096 --epi 4
"""
import argparse
import os
import sys

os.environ.setdefault("DLIA_GEMM_ROWS_DIAG", "1")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_llm_inference_amd import ops
from distributed_llm_inference_amd.ops import packing


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=28672)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--epi", type=int, default=4)
    a = ap.parse_args()
    assert a.m > 128, "the loader-wave kernel serves M > 128"
    K = ops.load()
    x = torch.randn(a.m, a.k, device="cuda").to(torch.bfloat16)
    wp = packing.pack_weight((torch.randn(a.n, a.k, device="cuda") * 0.02).to(torch.bfloat16))
    n_out = a.n // 2 if a.epi == 4 else a.n
    out = torch.empty(a.m, n_out, device="cuda", dtype=torch.bfloat16)
    wgs = ((a.n + 127) // 128 + 7) // 8 * 8 * ((a.m + 255) // 256)
    dbg = torch.zeros(wgs * 12 * 16, device="cuda", dtype=torch.float32)
    for _ in range(3):
        K.gemm(out, x, wp, None, None, dbg, a.n, a.epi, 1, 0)
    torch.cuda.synchronize()
    d = dbg.view(torch.int64).view(wgs, 12, 8).cpu().double()
    d = d[d[:, 0, 3] > 0]                          # workgroups that ran (the grid is padded)
    chunks = d[:, :, 3:4]
    per = d[:, :, :3] / chunks                     # cycles per chunk
    comp, load = per[:, :8].reshape(-1, 3), per[:, 8:].reshape(-1, 3)
    print(f"M={a.m} N={a.n} K={a.k}: {d.shape[0]} workgroups, {int(chunks.max())} chunks each; shader cycles per chunk, median (p90)")
    ghz = (d[:, :8, 4] / d[:, :8, 5].clamp_min(1)).reshape(-1) * 0.1
    print(f"  in-kernel clock over the chunk loop: median {ghz.median():.2f} GHz (p10 {ghz.quantile(0.1):.2f}, p90 {ghz.quantile(0.9):.2f})")
    for name, t, cols in (("compute wave", comp, ("barrier wait", "first fragment reads", "MFMA section (32 MFMAs + reads)")),
                          ("loader wave", load, ("vmcnt wait (data landing)", "barrier wait", "DMA issue (12 pieces)"))):
        tot = t.sum(1)
        print(f"  {name}: total {tot.median():7.0f}")
        for i, c in enumerate(cols):
            print(f"    {c:34s} {t[:, i].median():7.0f} ({t[:, i].quantile(0.9):7.0f})  {100 * t[:, i].median() / tot.median():5.1f} %")


if __name__ == "__main__":
    main()
