"""Single-GPU decode micro-benchmark of one pipeline stage (all layers): times graph-replayed
decode steps of one micro-batch. Used for kernel iteration and rocprofv3 captures."""
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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--ctx", type=int, default=128)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--layers", type=int, default=0, help="truncate the model (profiling)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--readback", type=int, default=0, metavar="DEPTH",
                    help="read every step's tokens back to the host the way the serving head does (side stream, pinned "
                         "buffer) and keep at most DEPTH steps in flight, as the scheduler does")
    ap.add_argument("--side-stream", action="store_true", help="run the steps on a non-default stream")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8", "fp8w"])
    ap.add_argument("--kv-dtype", default="bf16", choices=["bf16", "fp8"], help="fp8: e4m3 KV cache with per-(token, head) scales")
    ap.add_argument("--zero-operands", action="store_true",
                    help="diagnostic: the decoder layers' weights set to zero after packing (same instructions, same bytes moved, "
                         "next to no switching in the matrix cores) - how much of a step's time is the power limit")
    a = ap.parse_args()
    spec = ModelSpec.from_pretrained(a.model)
    if a.layers:
        import dataclasses
        spec = dataclasses.replace(spec, num_layers=a.layers)
    plan = make_plans(spec, 1)[0]
    t0 = time.time()
    w = random_stage_weights(spec, plan, seed=0, device="cuda", dtype=torch.bfloat16)
    B = a.batch
    max_len = a.ctx + a.steps + a.warmup + 8
    nblk = (max_len + 31) // 32
    st = GpuStage(spec, plan, w, max_slots=B, max_seq_len=nblk * 32, num_blocks=B * nblk, max_tokens=max(B, 64),
                  max_batch=B, use_graphs=not a.no_graph, weight_dtype=a.dtype, kv_dtype=a.kv_dtype)
    del w
    if a.zero_operands:
        for layer in st.layers:
            for t_ in layer.values():
                if torch.is_tensor(t_):
                    t_.zero_()
        # (embedding and lm_head stay: the hidden state is then the embedding all the way, the logits differ from
        # row to row and the sampler takes its usual path; with all-equal logits it walks whole rows)
    torch.cuda.synchronize()
    print(f"built in {time.time() - t0:.1f}s", st.memory_report(), flush=True)
    tables = {s: list(range(s * nblk, (s + 1) * nblk)) for s in range(B)}
    b = StepBatch(slots=list(range(B)), q_lens=[1] * B, ctx_lens=[a.ctx] * B, sample=[1] * B,
                  temperature=[0.7] * B, top_k=[50] * B, top_p=[0.9] * B, seeds=list(range(B)),
                  sample_steps=[0] * B, table_slots=list(tables), table_blocks=list(tables.values()),
                  token_ids=[1] * B)
    t = st.decode_step(b)
    b.replay = True
    b.token_ids = None
    for _ in range(a.warmup):
        t = st.decode_step(b, x=t)
    torch.cuda.synchronize()
    run_stream = torch.cuda.Stream() if a.side_stream else torch.cuda.current_stream()
    res_stream = torch.cuda.Stream()
    ring = [(torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros(B, dtype=torch.int32).pin_memory()) for _ in range(6)]
    inflight = []
    t1 = time.perf_counter()
    with torch.cuda.stream(run_stream):
        for i in range(a.steps):
            if a.readback:
                while len(inflight) >= a.readback:
                    if inflight[0].query():
                        inflight.pop(0)
                    else:
                        time.sleep(20e-6)
                t = st.decode_step(b)              # single stage: the graph feeds its own tokens back
                stg, host = ring[i % 6]
                stg.copy_(t)
                done, ev = torch.cuda.Event(), torch.cuda.Event()
                done.record()
                with torch.cuda.stream(res_stream):
                    res_stream.wait_event(done)
                    host.copy_(stg, non_blocking=True)
                    ev.record()
                inflight.append(ev)
            else:
                t = st.decode_step(b, x=t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / a.steps
    wbytes = st.memory_report()["weights_gib"] * (1 << 30)
    if spec.tie_embeddings is False and st.embed is not None:
        wbytes -= st.embed.numel() * 2
    print(json.dumps({"model": spec.name, "batch": B, "ctx": a.ctx, "ms_per_step": dt * 1e3,
                      "tok_per_s": B / dt, "weight_stream_TBps": wbytes / dt / 1e12}))


if __name__ == "__main__":
    main()
