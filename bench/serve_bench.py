"""Closed-loop serving benchmark of the whole engine in one process: R requests with random prompt
lengths arrive at once, the scheduler batches them continuously (chunked prefill + decode
micro-batches), and the script reports request throughput, output tokens/s, TTFT and TPOT.
Unlike bench.py (steady-state decode steps only) this includes prefill, admission, retirement and
the host-side scheduler."""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_llm_inference_amd.config import EngineConfig
from distributed_llm_inference_amd.parallel.launcher import make_pipeline
from distributed_llm_inference_amd.runtime.sampler import SamplingParams


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--requests", type=int, default=512)
    ap.add_argument("--prompt-min", type=int, default=32)
    ap.add_argument("--prompt-max", type=int, default=256)
    ap.add_argument("--new-tokens", type=int, default=64)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--micro-batches", type=int, default=2)
    ap.add_argument("--pp-size", type=int, default=1)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--kv-dtype", default="bf16", choices=["bf16", "fp8"])
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--transport", default="auto", help="EngineConfig.transport (loopback: all stages in this process, sharing the GPU)")
    ap.add_argument("--isolate", action="store_true", help="stage processes as children of this one (what `serve` does on GPUs)")
    ap.add_argument("--max-seq-len", type=int, default=0)
    ap.add_argument("--prefill-chunk", type=int, default=0, help="EngineConfig.prefill_chunk: tokens per prefill step")
    ap.add_argument("--closed-loop", type=int, default=0, metavar="CLIENTS",
                    help="CLIENTS requests outstanding at any time, each completion submits the next (default: all at t = 0)")
    ap.add_argument("--admit-fill", type=float, default=None, help="EngineConfig.admit_fill_fraction")
    ap.add_argument("--admit-wait-ms", type=float, default=None, help="EngineConfig.admit_max_wait_ms")
    ap.add_argument("--profile", action="store_true", help="cProfile the scheduler thread (in-process pipelines) and print the top of it to stderr")
    ap.add_argument("--trace", default=None, help="EngineConfig.trace_path: Chrome trace of host and GPU spans")
    ap.add_argument("--kv-blocks", type=int, default=None, help="EngineConfig.kv_blocks (default: sized from device memory)")
    ap.add_argument("--inflight", type=int, default=None, help="EngineConfig.max_inflight_steps")
    ap.add_argument("--mixed-prefill", type=int, default=None, choices=[0, 1],
                    help="EngineConfig.mixed_prefill: running sequences ride in their micro-batch's prefill steps (default: on)")
    ap.add_argument("--stagger-ms", type=float, default=0.0, help="closed loop: spread the first CLIENTS arrivals over this long")
    a = ap.parse_args()
    cfg = EngineConfig(model=a.model, pp_size=a.pp_size, max_batch=a.batch, num_micro_batches=a.micro_batches,
                       max_seq_len=a.max_seq_len or ((a.prompt_max + a.new_tokens + 31) // 32) * 32, prefill_chunk=a.prefill_chunk or EngineConfig().prefill_chunk, isolate=a.isolate, dtype=a.dtype, kv_dtype=a.kv_dtype,
                       device="cpu" if a.cpu else "auto", backend="torch" if a.cpu else "auto",
                       transport="loopback" if a.cpu else a.transport)
    if a.mixed_prefill is not None:
        cfg.mixed_prefill = bool(a.mixed_prefill)
    if a.admit_fill is not None:
        cfg.admit_fill_fraction = a.admit_fill
    if a.inflight is not None:
        cfg.max_inflight_steps = a.inflight
    if a.trace:
        cfg.trace_path = a.trace
    if a.kv_blocks is not None:
        cfg.kv_blocks = a.kv_blocks
    if a.admit_wait_ms is not None:
        cfg.admit_max_wait_ms = a.admit_wait_ms
    prof = None
    if a.profile:
        import cProfile
        from distributed_llm_inference_amd.runtime import engine as _eng
        prof = cProfile.Profile()
        inner = _eng.Engine._loop
        _eng.Engine._loop = lambda self: prof.runcall(inner, self)
    pipe = make_pipeline(cfg)
    eng = pipe.engine
    g = torch.Generator().manual_seed(0)
    V = eng.spec.vocab_size
    prompts = [torch.randint(10, V - 10, (int(torch.randint(a.prompt_min, a.prompt_max + 1, (1,), generator=g)),),
                             generator=g).tolist() for _ in range(a.requests)]
    sp = SamplingParams(temperature=0.7, top_k=50, top_p=0.9, max_new_tokens=a.new_tokens, stop_on_eos=False)
    try:
        warm = [eng.submit(p, SamplingParams(0.7, 50, 0.9, 4, stop_on_eos=False)) for p in prompts[:a.batch]]
        for r in warm:
            r.wait(600)
        t0 = time.perf_counter()
        if a.closed_loop:
            import threading
            lock, reqs, nxt, done = threading.Lock(), [], [0], threading.Event()

            def launch(_=None):
                with lock:
                    i = nxt[0]
                    if i >= len(prompts):
                        if all(r.finished or r.state == "failed" for r in reqs):
                            done.set()
                        return
                    nxt[0] += 1
                    r = eng.submit(prompts[i], sp)
                    reqs.append(r)
                r.add_done_callback(launch)

            for c in range(min(a.closed_loop, len(prompts))):
                launch()
                if a.stagger_ms:
                    time.sleep(a.stagger_ms / 1e3 / a.closed_loop)
            while not done.wait(0.05):
                with lock:
                    if nxt[0] >= len(prompts) and all(r.finished or r.state == "failed" for r in reqs):
                        break
        else:
            reqs = [eng.submit(p, sp) for p in prompts]
            for r in reqs:
                r.wait(1800)
        dt = time.perf_counter() - t0
    finally:
        snap = eng.snapshot()
        pipe.close()
    if prof is not None:
        import pstats
        import sys as _sys
        pstats.Stats(prof, stream=_sys.stderr).sort_stats("cumulative").print_stats(45)
    out_tokens = sum(len(r.output) for r in reqs)
    ttft = sorted((r.t_first - r.t_arrival) * 1e3 for r in reqs if r.t_first)
    tpot = sorted(((r.t_done - r.t_first) / max(1, len(r.output) - 1)) * 1e3 for r in reqs if r.t_first and r.t_done)
    # inter-token latency: every gap between two consecutive tokens of a request after its first (the
    # gap behind the first token is reported separately: a request whose prompt is done early waits there
    # until the rest of its micro-batch has been prefilled)
    itl = sorted((b - a) * 1e3 for r in reqs for a, b in zip(r.token_times[1:], r.token_times[2:]))
    second = sorted((r.token_times[1] - r.token_times[0]) * 1e3 for r in reqs if len(r.token_times) > 1)
    q = lambda v, p: v[min(len(v) - 1, int(p * len(v)))] if v else None
    print(json.dumps({"model": eng.spec.name, "requests": a.requests, "seconds": round(dt, 3),
                      "prompt_tokens": sum(len(p) for p in prompts), "output_tokens": out_tokens,
                      "output_tok_per_s": round(out_tokens / dt, 1), "requests_per_s": round(a.requests / dt, 2),
                      "ttft_ms_p50": q(ttft, 0.5), "ttft_ms_p99": q(ttft, 0.99), "tpot_ms_p50": q(tpot, 0.5),
                      "tpot_ms_p99": q(tpot, 0.99), "itl_ms_p50": q(itl, 0.5), "itl_ms_p99": q(itl, 0.99),
                      "first_to_second_token_ms_p50": q(second, 0.5), "recompositions": snap.get("recompositions"),
                      "config": {"max_batch": a.batch, "micro_batches": a.micro_batches, "pp": a.pp_size, "dtype": a.dtype, "kv_dtype": a.kv_dtype,
                                 "isolate": a.isolate, "closed_loop_clients": a.closed_loop, "stagger_ms": a.stagger_ms},
                      "steps": {k: snap.get(k) for k in ("prefill_steps", "decode_steps", "preemptions")}}))


if __name__ == "__main__":
    main()
