"""Sweep (n-tiles per wave, split-K) of the weight-streaming GEMM on the decode shapes of a model
and report achieved weight bandwidth per configuration, next to the heuristic's own pick.
All variants are timed interleaved in one process (cold caches: a large buffer is rewritten
between timings so weights come from HBM, as in a real decode step)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_llm_inference_amd import ops
from distributed_llm_inference_amd.models.spec import ModelSpec
from distributed_llm_inference_amd.ops import packing


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--m", type=int, nargs="+", default=[64])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--a8w8", action="store_true", help="sweep the fp8 x fp8 kernel instead of the bf16 one")
    a = ap.parse_args()
    K = ops.load()
    s = ModelSpec.from_pretrained(a.model)
    H, I = s.hidden_size, s.intermediate_size
    shapes = [("qkv", s.qkv_size, H, 6), ("o", H, s.q_size, 6), ("gate_up", 2 * I, H, 4), ("down", H, I, 6),
              ("lm_head", packing.padded_n(s.vocab_size), H, 0)]
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    for M in a.m:
        for name, N, Kd, epi in shapes:
            x = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
            wraw = torch.randn(N, Kd, device="cuda") * 0.02
            wp = packing.pack_weight(wraw.to(torch.bfloat16))
            if a.a8w8:
                wq, wsc = packing.pack_weight_a8w8(wraw)
                xq = torch.zeros(M, Kd, device="cuda", dtype=torch.uint8)
                xs = torch.zeros(M, device="cuda")
                K.quant_rows(xq, xs, x)
            n_out = N // 2 if epi == 4 else N
            out = torch.empty(M, n_out, device="cuda", dtype=torch.float32 if name == "lm_head" else torch.bfloat16)
            ws = torch.empty(16 * M * N, device="cuda", dtype=torch.float32)
            auto = tuple(int(v) for v in K.gemm_config(M, N, Kd, epi, True))
            cands = [(n, sk) for n in (1, 2) for sk in (1, 2, 4, 8, 16) if not (epi == 4 and n == 1) and Kd // 128 // sk >= 2]
            res = {}
            for (n, sk) in cands:
                ev = []
                for it in range(a.iters + 2):
                    flush.fill_(it & 1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    if a.a8w8:
                        K.gemm_a8w8(out, xq, xs, wq, wsc, None, None, ws, N, epi, sk, n)
                    else:
                        K.gemm(out, x, wp, None, None, ws, N, epi, sk, n)
                    e1.record()
                    ev.append((e0, e1))
                torch.cuda.synchronize()
                ts = sorted(a_.elapsed_time(b_) * 1e3 for a_, b_ in ev[2:])
                res[(n, sk)] = ts[len(ts) // 2]
            best = min(res, key=res.get)
            gb = N * Kd * (1 if a.a8w8 else 2) / 1e3
            line = " ".join(f"{n}/{sk}:{t:5.1f}" for (n, sk), t in sorted(res.items()))
            print(f"M={M:3d} {name:8s} N={N:6d} K={Kd:5d} | auto {auto} {res.get(auto, float('nan')):5.1f}us "
                  f"({gb / res.get(auto, 1e9):.0f} GB/s) | best {best} {res[best]:5.1f}us ({gb / res[best]:.0f} GB/s) | {line}",
                  flush=True)


if __name__ == "__main__":
    main()
