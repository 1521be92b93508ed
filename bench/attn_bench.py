#!/usr/bin/env python3
"""Decode attention and RoPE + KV write alone, at the shapes of one decode step: B sequences with
`ctx` cached tokens each, one new token per sequence. Reports time per call and the HBM rate over
the K/V bytes a step has to read (whole 32-token blocks), for A/B runs of kernel variants
(DLIA_ATTN_VARIANT, DLIA_ATTN_WAVE_ITEMS are read when the extension first launches the kernel).

    python bench/attn_bench.py --batch 1024 --ctx 150
    python bench/attn_bench.py --batch 1024 --ctx 150 --kv-dtype fp8
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_llm_inference_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--ctx", type=int, default=150, help="cached tokens per sequence before this step")
    ap.add_argument("--jitter", type=int, default=0, help="contexts drawn from [ctx - jitter, ctx + jitter]")
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--kv-dtype", default="bf16", choices=["bf16", "fp8"])
    ap.add_argument("--layers", type=int, default=8, help="distinct caches cycled through, so no call finds its K/V in L2 / MALL")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--heat", action="store_true",
                    help="run a large GEMM between the timed calls and time each call with its own events: the clocks "
                         "a decode step's GEMMs leave behind, not the idle-chip boost clock")
    ap.add_argument("--scatter", action="store_true", help="random block order instead of consecutive blocks per sequence")
    a = ap.parse_args()
    K = ops.load()
    dev = "cuda"
    B, Hq, Hkv, D = a.batch, a.hq, a.hkv, a.d
    kv8 = a.kv_dtype == "fp8"
    g = torch.Generator().manual_seed(0)
    ctx = torch.full((B,), a.ctx, dtype=torch.int32)
    if a.jitter:
        ctx = ctx + torch.randint(-a.jitter, a.jitter + 1, (B,), generator=g, dtype=torch.int32)
    max_blocks = (int(ctx.max()) + 1 + 31) // 32
    num_blocks = B * max_blocks
    order = torch.randperm(num_blocks, generator=g) if a.scatter else torch.arange(num_blocks)
    table = order.reshape(B, max_blocks).to(device=dev, dtype=torch.int32)
    kvs = [torch.randn(num_blocks, Hkv, 2, 32 * D, device=dev).mul_(0.5).to(torch.float8_e4m3fn).view(torch.uint8) if kv8
           else torch.randn(num_blocks, Hkv, 2, 32 * D, device=dev, dtype=torch.bfloat16) for _ in range(a.layers)]
    scales = [torch.rand(num_blocks, Hkv, 2, 32, device=dev) + 0.5 for _ in range(a.layers)] if kv8 else None
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device=dev, dtype=torch.bfloat16)
    out = torch.empty(B, Hq * D, device=dev, dtype=torch.bfloat16)
    q_start = torch.arange(B + 1, device=dev, dtype=torch.int32)
    ctx_d = ctx.to(dev)
    pos = ctx_d.clone()
    seq = torch.arange(B, device=dev, dtype=torch.int32)
    half = D // 2
    ang = torch.arange(8192, device=dev)[:, None] * (1.0 / (500000.0 ** (torch.arange(half, device=dev) / half)))[None, :]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    scale = 1.0 / math.sqrt(D)

    def attn(i):
        if kv8:
            K.paged_attention8(out, qkv, kvs[i], scales[i], q_start, ctx_d, table, None, None, 1, Hq, Hkv, D, scale, 1)
        else:
            K.paged_attention(out, qkv, kvs[i], q_start, ctx_d, table, None, None, 1, Hq, Hkv, D, scale, 1)

    def write(i):
        if kv8:
            K.rope_kvwrite8(qkv, None, 1, None, kvs[i], scales[i], cos, sin, pos, seq, table, Hq, Hkv, D, True)
        else:
            K.rope_kvwrite(qkv, None, 1, None, kvs[i], cos, sin, pos, seq, table, Hq, Hkv, D, True)

    def timed(fn):
        for i in range(a.layers):
            fn(i)
        torch.cuda.synchronize()
        if a.heat:
            ha = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
            hb = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
            evs = []
            for it in range(a.iters):
                for _ in range(2):
                    torch.matmul(ha, hb)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(it % a.layers)
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ts = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
            return ts[len(ts) // 2]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(a.iters):
            fn(it % a.layers)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.iters

    t_attn = timed(attn)
    t_write = timed(write)
    blocks = ((ctx + 1 + 31) // 32).sum().item()                   # the new token's K/V is read back too
    bytes_kv = blocks * Hkv * 2 * 32 * D * (1 if kv8 else 2) + (blocks * Hkv * 2 * 32 * 4 if kv8 else 0)
    bytes_qo = B * Hq * D * 2 * 2
    print(json.dumps({"batch": B, "ctx": a.ctx, "kv_dtype": a.kv_dtype, "attention_us": round(t_attn, 1),
                      "kv_TBps": round(bytes_kv / t_attn / 1e6, 2), "kv_plus_q_out_TBps": round((bytes_kv + bytes_qo) / t_attn / 1e6, 2),
                      "rope_kvwrite_us": round(t_write, 1), "variant": os.environ.get("DLIA_ATTN_VARIANT", "default"),
                      "wave_items": os.environ.get("DLIA_ATTN_WAVE_ITEMS", "default"), "scatter": a.scatter}))


if __name__ == "__main__":
    main()
