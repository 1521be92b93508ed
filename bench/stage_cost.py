"""What ONE pipeline stage costs per decode step in the benchmark's regime: micro-batches of 2048
sequences alternating over two HIP streams, hipGraph decode, ~150 cached tokens — for any stage
shape, given as block ranges (block 2 i = attention of layer i, 2 i + 1 = its MLP) with or without
the embedding and the head. This is what the partition cost model (models/partition.py:
ATTN_BLOCK_SHARE, HEAD_TIME_PER_PARAM) has to predict: on a full node every stage runs in exactly this
regime, and the slowest one sets the step time.

    python bench/stage_cost.py --stages 0-8 0-8+embed 58-64+head 0-9 1-9
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_llm_inference_amd.models.partition import StagePlan
from distributed_llm_inference_amd.models.spec import ModelSpec
from distributed_llm_inference_amd.models.stage_gpu import GpuStage
from distributed_llm_inference_amd.models.weights import random_stage_weights
from distributed_llm_inference_amd.runtime.batch import StepBatch


def plan_of(text: str, spec) -> StagePlan:
    parts = text.split("+")
    a, b = (int(v) for v in parts[0].split("-"))
    return StagePlan(stage=1, num_stages=3, layer_start=a // 2, layer_end=(b + 1) // 2, has_embed="embed" in parts[1:],
                     has_head="head" in parts[1:], skip_attn_first=(a % 2 == 1), attn_only_last=(b % 2 == 1))


def measure(spec, plan, a):
    B, nmb = a.batch, a.micro_batches
    w = random_stage_weights(spec, plan, seed=0, device="cuda", dtype=torch.bfloat16)
    max_len = a.ctx + a.steps + a.warmup + 8
    nblk = (max_len + 31) // 32
    st = GpuStage(spec, plan, w, max_slots=nmb * B, max_seq_len=nblk * 32, num_blocks=nmb * B * nblk, max_tokens=max(B, 64),
                  max_batch=B, use_graphs=True, weight_dtype=a.dtype, kv_dtype=a.kv_dtype, decode_states=nmb)
    del w
    streams = [torch.cuda.Stream() for _ in range(a.streams)]
    descs = []
    for mb in range(nmb):
        slots = [mb * B + i for i in range(B)]
        d = StepBatch(micro_batch=mb, slots=slots, q_lens=[1] * B, ctx_lens=[a.ctx] * B, sample=[1] * B,
                      temperature=[0.7] * B, top_k=[50] * B, top_p=[0.9] * B, seeds=slots, sample_steps=[0] * B,
                      table_slots=slots, table_blocks=[list(range(s * nblk, (s + 1) * nblk)) for s in slots],
                      token_ids=[1 + (s % 1000) for s in slots])
        x = None if plan.has_embed else torch.randn(B, spec.hidden_size, device="cuda").to(torch.bfloat16)
        with torch.cuda.stream(streams[mb % len(streams)]):
            st.decode_step(d, x=x)
        d.replay, d.token_ids = True, None
        descs.append(d)
    torch.cuda.synchronize()

    def rounds(n):
        for _ in range(n):
            for mb in range(nmb):
                with torch.cuda.stream(streams[mb % len(streams)]):
                    st.decode_step(descs[mb])
    rounds(a.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rounds(a.steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    del st
    torch.cuda.empty_cache()
    return dt * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--stages", nargs="+", default=["0-8", "0-8+embed", "58-64+head", "0-9", "1-9", "0-4", "0-16"])
    ap.add_argument("--plan", type=int, default=0, metavar="N",
                    help="measure the stages of the balanced N-stage partition instead of --stages, and print the scaling bound they give")
    ap.add_argument("--whole-layers", action="store_true")
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--micro-batches", type=int, default=2)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--ctx", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--kv-dtype", default="bf16")
    a = ap.parse_args()
    spec = ModelSpec.from_pretrained(a.model)
    out = {}
    if a.plan:
        from distributed_llm_inference_amd.models.partition import make_plans, stage_costs
        plans = make_plans(spec, a.plan, split_layers=not a.whole_layers, rows=a.batch)
        times = []
        for p, c in zip(plans, stage_costs(spec, plans, a.batch)):
            ms = measure(spec, p, a)
            times.append(ms)
            print(f"stage {p.stage} {p.describe():10s}{' + head' if p.has_head else ''}: {ms:8.3f} ms per step of {a.micro_batches} x {a.batch} rows "
                  f"(cost model: {c:.2f} layers)", flush=True)
        print(json.dumps({"stages": a.plan, "ms": [round(t, 3) for t in times],
                          "bound_mean_over_max": round(sum(times) / (len(times) * max(times)), 4)}))
        return
    for text in a.stages:
        ms = measure(spec, plan_of(text, spec), a)
        out[text] = round(ms, 3)
        print(f"stage {text:16s}: {ms:8.3f} ms per step of {a.micro_batches} x {a.batch} rows", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
