"""Prefill-regime GEMM micro-benchmark: the LDS-tiled MFMA kernel on the Llama-3-8B layer shapes."""
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
    ap.add_argument("--m", type=int, nargs="+", default=[512, 2048, 8192])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-split", action="store_true", help="no workspace: the 256 x 256 kernel cannot split K inside the launch")
    ap.add_argument("--a8w8", action="store_true", help="the fp8 x fp8 kernel (e4m3 operands, MX MFMA) instead of bf16")
    a = ap.parse_args()
    K = ops.load()
    s = ModelSpec.from_pretrained(a.model)
    H, I = s.hidden_size, s.intermediate_size
    shapes = [("qkv", s.qkv_size, H, 0), ("o", H, s.q_size, 2), ("gate_up", 2 * I, H, 4), ("down", H, I, 2)]
    for M in a.m:
        tot_t, tot_f = 0.0, 0.0
        for name, N, Kd, epi in shapes:
            x = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
            wp = packing.pack_weight((torch.randn(N, Kd, device="cuda") * 0.02).to(torch.bfloat16))
            n_out = N // 2 if epi == 4 else N
            out = torch.empty(M, n_out, device="cuda", dtype=torch.bfloat16)
            res = torch.randn(M, n_out, device="cuda").to(torch.bfloat16) if epi == 2 else None
            need = 0 if a.no_split else int(K.gemm_ws_floats(M, N, Kd))
            ws = torch.zeros(need + 4096, device="cuda") if need else None
            if a.a8w8:
                wq, wsc = packing.pack_weight_a8w8(torch.randn(N, Kd, device="cuda") * 0.02)
                xq = torch.zeros(M, Kd, device="cuda", dtype=torch.uint8)
                xs = torch.zeros(M, device="cuda")
                K.quant_rows(xq, xs, x)

            def run():
                if a.a8w8:
                    K.gemm_a8w8(out, xq, xs, wq, wsc, None, res, ws, N, epi, 1, 0)
                else:
                    K.gemm(out, x, wp, None, res, ws, N, epi, 1, 0)
            for _ in range(2):
                run()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(a.iters):
                run()
            ev[1].record()
            torch.cuda.synchronize()
            t = ev[0].elapsed_time(ev[1]) / a.iters * 1e-3
            fl = 2.0 * M * N * Kd
            tot_t += t; tot_f += fl
            print(f"M={M:5d} {name:8s} N={N:6d} K={Kd:6d}: {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s")
        print(f"M={M:5d} layer total: {tot_t * 1e6:8.1f} us  {tot_f / tot_t / 1e12:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
