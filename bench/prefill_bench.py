"""Single-GPU prefill micro-benchmark of one pipeline stage: time to run a chunk of T prompt
tokens (B sequences of T/B tokens) through all layers, eager path (prefill is not graph-captured)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_llm_inference_amd.models.partition import make_plans
from distributed_llm_inference_amd.models.spec import ModelSpec
from distributed_llm_inference_amd.models.stage_gpu import GpuStage
from distributed_llm_inference_amd.models.weights import random_stage_weights
from distributed_llm_inference_amd.runtime.batch import StepBatch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--tokens", type=int, default=2048)
    ap.add_argument("--seqs", type=int, default=4)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8", "fp8w"])
    ap.add_argument("--kv-dtype", default="bf16", choices=["bf16", "fp8"], help="fp8: e4m3 KV cache with per-(token, head) scales")
    a = ap.parse_args()
    spec = ModelSpec.from_pretrained(a.model)
    if a.layers:
        import dataclasses
        spec = dataclasses.replace(spec, num_layers=a.layers)
    plan = make_plans(spec, 1)[0]
    w = random_stage_weights(spec, plan, seed=0, device="cuda", dtype=torch.bfloat16)
    B, T = a.seqs, a.tokens
    per = T // B
    nblk = (per + 31) // 32 + 1
    st = GpuStage(spec, plan, w, max_slots=B, max_seq_len=nblk * 32, num_blocks=B * nblk, max_tokens=T, max_batch=max(B, 8),
                  use_graphs=False, weight_dtype=a.dtype, kv_dtype=a.kv_dtype)
    del w
    tables = {s: list(range(s * nblk, (s + 1) * nblk)) for s in range(B)}
    ids = torch.randint(10, spec.vocab_size - 10, (per * B,), dtype=torch.int32)
    d = StepBatch(slots=list(range(B)), q_lens=[per] * B, ctx_lens=[0] * B, sample=[1] * B, temperature=[0.0] * B,
                  top_k=[50] * B, top_p=[0.9] * B, seeds=[0] * B, sample_steps=[0] * B, table_slots=list(tables),
                  table_blocks=list(tables.values()), token_ids=ids.tolist())
    st.forward(d, ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        st.forward(d, ids)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    flops = 2.0 * per * B * (spec.layer_param_count() * spec.num_layers)
    print(json.dumps({"model": spec.name, "tokens": per * B, "seqs": B, "ms": dt * 1e3, "prefill_tok_per_s": per * B / dt,
                      "gemm_tflops": flops / dt / 1e12, "dtype": a.dtype}))


if __name__ == "__main__":
    main()
