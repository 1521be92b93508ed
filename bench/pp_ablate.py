"""Ablation timing of the 256 x 256 ping-pong GEMM: diagnostic builds of gemm_pingpong.hip with parts
of the main loop left out (PP_ABL mask: 1 DMA issue, 2 fragment reads, 4 barriers, 8 MFMAs, 2048 three quarters of the epilogue; for the four-wave
kernel, selected with DLIA_GEMM_LARGE=quad: 64 landing waits, 128 barriers, 256 DMA issue, 512 all but one row of the
epilogue), timed on
the Llama-3-8B layer shapes. The outputs of an ablated build are wrong on purpose; the times say what
the remaining parts cost. Build on the CPU box first (`--build`), run on the GPU.

    python bench/pp_ablate.py --build            # compiles bench/build/pp_abl_<mask>.so for every mask
    python bench/pp_ablate.py --m 4096
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "distributed_llm_inference_amd", "ops", "csrc")
OUT = os.path.join(HERE, "build")
MASKS = [0, 1, 2, 3, 4, 8, 9, 10, 12]


def so_path(mask):
    return os.path.join(OUT, f"pp_abl_{mask}.so")


def build(masks):
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for m in masks:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
               f"-DPP_ABL={m}", os.path.join(CSRC, "gemm_pingpong.hip"), os.path.join(CSRC, "gemm_rows.hip"),
               "-o", so_path(m)]
        procs.append((m, subprocess.Popen(cmd)))
        if len(procs) >= 4:
            for mm, p in procs:
                assert p.wait() == 0, mm
            procs = []
    for mm, p in procs:
        assert p.wait() == 0, mm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--m", type=int, nargs="+", default=[4096])
    ap.add_argument("--masks", type=int, nargs="+", default=MASKS)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--pad", type=int, nargs="+", default=[0], help="extra elements per x row (row stride K + pad): L2 channel spread")
    a = ap.parse_args()
    if a.build:
        build(a.masks)
        return
    import torch
    sys.path.insert(0, ROOT)
    from distributed_llm_inference_amd.ops import packing
    shapes = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]
    vp = C.c_void_p
    for M, pad in [(m, p) for m in a.m for p in a.pad]:
        data = {}
        for name, N, K in shapes:
            x = torch.randn(M, K + pad, device="cuda").to(torch.bfloat16)
            wp = packing.pack_weight((torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16))
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            data[name] = (x, wp, out)
        for mask in a.masks:
            L = C.CDLL(so_path(mask))
            f = L.dlia_gemm_large
            f.restype = None
            # (out, x, wp, bias, residual, workspace, ws_floats, M, N, K, ldx, ldo, ldr, epi, out_f32, wg_target, stream,
            #  chunk_max, ldc, n_valid): every argument of the C signature is listed — an argument left out is read as
            # whatever the register holds, and a stray `chunk_max` pointer is written through
            f.argtypes = [vp, vp, vp, vp, vp, vp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long,
                          C.c_int, C.c_int, C.c_int, vp, vp, C.c_long, C.c_int]
            line = [f"M={M} pad={pad} mask={mask:2d} (" + "+".join(n for b, n in ((1, "noDMA"), (2, "noREAD"), (4, "noBAR"), (8, "noMFMA"), (16, "sameTile"), (32, "stamps"), (64, "q:noWAIT"), (128, "q:noBAR"), (256, "q:noDMA"), (512, "q:noEPILOGUE"), (1024, "q:stamps"), (2048, "quarterEPILOGUE")) if mask & b) + ")"]
            tt, tf = 0.0, 0.0
            for name, N, K in shapes:
                x, wp, out = data[name]
                st = torch.cuda.current_stream().cuda_stream

                stamps = torch.zeros(((M + 255) // 256) * ((N + 2047) // 2048 * 8) * 8 * 16 + 64, device="cuda", dtype=torch.int32) if mask & (32 | 1024) else None

                def run():
                    f(out.data_ptr(), x.data_ptr(), wp.data_ptr(), None, None, stamps.data_ptr() if stamps is not None else None,
                      0, M, N, K, K + pad, N, 0, 0, 0, 0, vp(st), None, 0, 0)
                for _ in range(3):
                    run()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                for _ in range(a.iters):
                    run()
                ev[1].record()
                torch.cuda.synchronize()
                t = ev[0].elapsed_time(ev[1]) / a.iters * 1e-3
                fl = 2.0 * M * N * K
                tt += t; tf += fl
                line.append(f"{name} {t * 1e6:7.1f}us {fl / t / 1e12:6.0f}T")
                if stamps is not None and mask & 1024:
                    v = stamps[:-64].view(-1, 8, 16)[:, :4, :8].float()
                    v = v[v[:, 0, 3] > 0]                           # workgroups that ran
                    c = v.mean(dim=(0, 1))                          # s_memtime ticks
                    print(f"    {name}: per workgroup, ticks: prologue {c[0].item():.0f}  K loop {c[1].item():.0f}  "
                          f"tail + epilogue {c[2].item():.0f} = DMA tail + barrier {c[4].item():.0f} + tiles {c[5].item():.0f} + "
                          f"store drain {c[6].item():.0f}  ({v.shape[0]} workgroups, {K // 64} K-tiles)", flush=True)
                elif stamps is not None:
                    v = stamps[:-64].view(-1, 8, 4, 4).float()
                    v = v[v.sum(dim=(1, 2, 3)) > 0]                 # workgroups that ran (the grid is padded)
                    ktiles = K // 64
                    for grp, name_g in ((slice(0, 4), "waves 0-3"), (slice(4, 8), "waves 4-7")):
                        m = v[:, grp].mean(dim=(0, 1)) / ktiles      # cycles per phase, [kind][segment]
                        print(f"    {name} {name_g}: cycles per phase (reads+DMA | barrier 1 | cluster | barrier 2): " +
                              "  ".join(f"q{q}: " + " | ".join(f"{m[q, sgm].item():5.0f}" for sgm in range(4)) for q in range(4)) +
                              f"   K-tile total {m.sum().item():6.0f}", flush=True)
            line.append(f"layer {tt * 1e6:7.1f}us {tf / tt / 1e12:6.0f}T")
            print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
