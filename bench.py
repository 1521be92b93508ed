#!/usr/bin/env python3
"""Headline benchmark: output tokens/sec (whole node) and p50 per-token latency of Llama-3-8B
layer-split over N pipeline stages (PP = N GPUs of one node), bf16, synthetic prompts,
random-init weights.

    python bench.py --gpus 1 --steps 32 --warmup 4
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One rank per GPU; rank = pipeline stage; hidden states hop stage to stage over RCCL send/recv,
sampled token ids hop back from the last stage to stage 0. A "step" is one decode step of EVERY
in-flight sequence: each of the micro-batches (2048 sequences each; 2 on one GPU, 2N + 2 on N > 1:
two per stage plus two spares that cover the hops) goes once through all N stages — embedding,
every decoder layer against the paged KV cache, final norm, lm_head, temperature/top-k/top-p
sampling, token feedback. Per-GPU work per token is constant in N (L/N layers per stage visit),
i.e. weak scaling. A stage rotates its micro-batches over 2 HIP
streams, so the kernels of two micro-batches overlap on each GPU (measured on one MI355X:
2 x 2048 rows 67.1k tok/s at 61 ms per token, 2 x 1536 65.5k at 47 ms, 2 x 1024 61.5k at 33 ms —
pick the latency point with --batch / --micro-batches / --decode-streams). Prefill of the synthetic prompts and W warm-up
steps run first; then exactly K steps are timed between barrier + device-synchronize brackets,
the MAX over ranks is taken, and rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# A stage process drives up to six HIP streams (eager, two decode streams, token read-back, and one per
# hop direction); with HIP's default of 4 hardware queues two of them would share a queue, and a
# receive waiting for its peer at the head of a shared queue stalls the compute behind it — the
# pipeline would run one micro-batch at a time. Read by the HIP runtime when it initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# RCCL and cross-process device buffers use dmabuf IPC handles; the legacy IPC mode is not supported by the driver here
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from distributed_llm_inference_amd.config import EngineConfig                        # noqa: E402
from distributed_llm_inference_amd.parallel.launcher import attach_distributed       # noqa: E402
from distributed_llm_inference_amd.parallel.pipeline import stage_main               # noqa: E402
from distributed_llm_inference_amd.runtime.batch import (KIND_BARRIER, KIND_SHUTDOWN,  # noqa: E402
                                                         StepBatch, control)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--batch", type=int, default=0,
                    help="sequences per micro-batch (default: 2048 for bf16 and fp8; 256 for fp8w, whose GEMM\n"
                         "streams at most 256 rows per pass)")
    ap.add_argument("--micro-batches", type=int, default=0,
                    help="in-flight micro-batches (default for bf16 and fp8: 2 on one GPU, 2 x gpus + 2 on more; 3 x gpus for fp8w)")
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--partition", default="balanced")
    ap.add_argument("--whole-layers", action="store_true", help="balanced partition over whole layers only (A/B)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--decode-streams", type=int, default=0,
                    help="HIP streams the decode micro-batches of a stage alternate over")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8", "fp8w"],
                    help="fp8: decoder-layer GEMMs on the fp8 MFMA (e4m3 weights + per-row quantised activations); "
                         "fp8w: e4m3 weights only; KV cache and lm_head stay bf16 in both")
    ap.add_argument("--kv-dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: e4m3 KV cache with per-(token, head) scales (half the attention bytes; not the default)")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "rccl-torch", "gloo"],
                    help="activation transport between stages (auto / rccl: native ncclSend/ncclRecv hops, one "
                         "communicator and stream per hop; rccl-torch: torch.distributed point-to-point; gloo = "
                         "host-staged TCP, which also lets several stages share one GPU for a rehearsal)")
    ap.add_argument("--cpu", action="store_true", help="plumbing run on CPU (torch backend, gloo)")
    ap.add_argument("--max-seconds", type=float, default=1500.0,
                    help="abort the whole rank with an error instead of hanging if the run takes longer")
    return ap.parse_args()


def metric_model(name: str) -> str:
    """'meta-llama/Meta-Llama-3-8B' -> 'Llama-3-8B' (the headline metric names the model)."""
    short = name.split("/")[-1]
    return short[5:] if short.startswith("Meta-") else short


def baseline_for(model: str):
    """Throughput (tok/s) of the reference's own CPU / JSON-over-HTTP path for this model, as measured
    by bench/reference_baseline.py and stored in bench/baseline_numbers.json; None if not measured."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench", "baseline_numbers.json")
    try:
        with open(path) as f:
            return json.load(f).get(model, {}).get("tokens_per_sec")
    except Exception:
        return None


class Driver:
    """Rank 0: issues the static benchmark schedule through its own StageRuntime."""

    def __init__(self, rt, cfg, args):
        self.rt, self.cfg, self.a = rt, cfg, args
        self.links = rt.links
        self.B = cfg.max_batch
        self.nmb = cfg.num_micro_batches
        self.on_gpu = rt.on_gpu
        self.res_stream = torch.cuda.Stream(device=rt.device) if self.on_gpu else None
        self.fb = [None] * self.nmb
        self.step_id = 0
        self.lat_events = [[] for _ in range(self.nmb)]
        self.blocks_per_seq = (cfg.max_seq_len + 31) // 32
        self._probe = StepBatch(slots=list(range(self.B)), q_lens=[1] * self.B)    # "a whole micro-batch decoding"

    def _desc(self, mb, q_lens, ctx_lens, sample, token_ids, replay=False, tables=False):
        B = self.B
        slots = [mb * B + i for i in range(B)]
        self.step_id += 1
        d = StepBatch(step_id=self.step_id, micro_batch=mb, slots=slots[:len(q_lens)], q_lens=q_lens,
                      ctx_lens=ctx_lens, sample=sample, temperature=[0.7] * len(q_lens), top_k=[50] * len(q_lens),
                      top_p=[0.9] * len(q_lens), seeds=[1000 + s for s in slots[:len(q_lens)]],
                      sample_steps=[0] * len(q_lens), token_ids=token_ids, replay=replay)
        return d

    def prefill(self):
        a, B = self.a, self.B
        g = torch.Generator().manual_seed(0)
        vocab = self.rt.ex.spec.vocab_size
        per_step = max(1, self.cfg.prefill_chunk // a.prompt_len)
        for mb in range(self.nmb):
            for s0 in range(0, B, per_step):
                n = min(per_step, B - s0)
                slots = [mb * B + s0 + i for i in range(n)]
                self.step_id += 1
                toks = torch.randint(10, vocab - 10, (n * a.prompt_len,), generator=g).tolist()
                d = StepBatch(step_id=self.step_id, micro_batch=mb, slots=slots, q_lens=[a.prompt_len] * n,
                              ctx_lens=[0] * n, sample=[0] * n, temperature=[0.7] * n, top_k=[50] * n,
                              top_p=[0.9] * n, seeds=slots, sample_steps=[0] * n, token_ids=toks,
                              table_slots=slots,
                              table_blocks=[list(range(s * self.blocks_per_seq, (s + 1) * self.blocks_per_seq))
                                            for s in slots])
                self.rt.execute(d)

    def decode_round(self, first: bool, ctx: int, record: bool = False):
        B = self.B
        for mb in range(self.nmb):
            if first:
                span = max(1, min(1000, self.rt.ex.spec.vocab_size - 20))
                d = self._desc(mb, [1] * B, [ctx] * B, [1] * B, [(7 * (mb * B + i)) % span + 10 for i in range(B)])
                fbw = None
            elif self.rt._graph_decode(self._probe):
                self.step_id += 1
                d = StepBatch.replay_of(self.step_id, mb, B, B)          # header-only descriptor: the stages replay their graphs
                fbw = self.fb[mb]
            else:
                d = self._desc(mb, [1] * B, [ctx] * B, [1] * B, None, replay=True)
                fbw = self.fb[mb]
            with self.rt.stream_scope(d):
                toks = self.rt.execute(d, fbw)
                if self.links.num_stages == 1 and self.on_gpu and record:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    self.lat_events[mb].append(ev)
            n = self.rt.result_count(d)
            if self.links.num_stages > 1:
                buf = self.rt.fb_buffer(mb, n)
                if self.on_gpu:
                    got = torch.cuda.Event(enable_timing=record)
                    with torch.cuda.stream(self.res_stream):
                        self.links.fb_data.recv(buf).wait()
                        got.record()
                    self.fb[mb] = _EvHandle(got)
                    if record:
                        self.lat_events[mb].append(got)
                else:
                    self.fb[mb] = self.links.fb_data.recv(buf)
                    if record:
                        self.fb[mb].wait()
                        self.lat_events[mb].append(time.perf_counter())
            elif record and not self.on_gpu:
                self.lat_events[mb].append(time.perf_counter())

    def barrier(self):
        self.rt.execute(control(KIND_BARRIER))

    def latencies_ms(self):
        out = []
        for evs in self.lat_events:
            for a, b in zip(evs, evs[1:]):
                out.append(a.elapsed_time(b) if hasattr(a, "elapsed_time") else (b - a) * 1e3)
        return sorted(out)


class _EvHandle:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)

    def done(self):
        return self.ev.query()


def _deadline(seconds: float):
    """A pipeline whose peer died would wait in a receive forever; fail loudly instead."""
    import threading

    def fire():
        sys.stderr.write(f"bench.py: no result after {seconds:.0f} s (rank {os.environ.get('RANK', '0')}), aborting\n")
        sys.stderr.flush()
        os._exit(3)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def _fit_batch(a, nmb: int, max_seq: int, world: int) -> int:
    """The KV cache holds every in-flight sequence for prompt + warm-up + timed steps. A long run
    (--steps in the hundreds) at 2 x 1024 sequences per GPU, or a 70B model on one GPU, would ask
    for more than the device has: halve the micro-batch until the largest stage's share fits in 85 %
    of what its weights leave, and say so (the size used is reported in the JSON line's config)."""
    from distributed_llm_inference_amd.parallel.launcher import plans_for, resolve_spec
    cfg = EngineConfig(model=a.model, pp_size=world, partition=a.partition, split_layers=not getattr(a, "whole_layers", False))
    spec = resolve_spec(cfg)
    plans = plans_for(cfg, spec)
    layers = max(len(p.attn_layers) for p in plans)           # one KV cache per attention block of a stage
    wbytes = 1 if a.dtype in ("fp8", "fp8w") else 2
    weights = max((p.layer_end - p.layer_start) * spec.layer_param_count()
                  + (spec.embed_param_count() if p.has_embed else 0) + (spec.head_param_count() if p.has_head else 0)
                  for p in plans) * wbytes
    idx = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    budget = 0.85 * (torch.cuda.get_device_properties(idx).total_memory - weights - (8 << 30))
    batch = a.batch
    while batch > 64 and nmb * batch * max_seq * spec.kv_bytes_per_token_per_layer(1 if a.kv_dtype == "fp8" else 2) * layers > budget:
        batch //= 2
    if batch != a.batch and int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write(f"bench.py: KV cache for {nmb} x {a.batch} sequences of {max_seq} tokens does not fit; "
                         f"using micro-batches of {batch}\n")
    return batch


def main():
    a = parse()
    timer = _deadline(a.max_seconds)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # operating point per dtype (measured on one MI355X): bf16 and fp8 2 x 2048 rows on 2 streams
    # (at 2048 rows the N = 4096 projections have 128 output tiles of 256 x 256 and need no split-K;
    # same box, bf16: 2 x 1024 61.5 k tok/s at 33.3 ms per step, 2 x 1536 65.5 k at 46.9 ms, 2 x 2048
    # 67.1 k at 61.1 ms; `--batch 1024` is the lower-latency point); the fp8-weights-only GEMM streams
    # at most 256 rows per pass, so larger micro-batches would re-read its weights
    wide = a.dtype in ("bf16", "fp8")
    a.batch = a.batch or (8 if a.cpu else (2048 if wide else 256))
    a.decode_streams = a.decode_streams or (2 if wide else 3)
    # More than one stage: two spare micro-batches. With exactly two per stage every micro-batch is always
    # being computed somewhere, so each hop's latency (16.8 MB of hidden state over one xGMI link, ~0.35 ms
    # against ~7.7 ms per stage visit at 8 stages) is a bubble nobody fills: bench/pipeline_sim.py gives 0.967
    # of the stages' capacity at 8 stages (0.90 with 1 ms hops) and 1.00 with two spares, whatever the hop.
    nmb = a.micro_batches or ((2 if wide else 3) * max(1, world) + (2 if world > 1 and wide else 0))
    max_seq = ((a.prompt_len + a.steps + a.warmup + 2 + 31) // 32) * 32
    if not a.cpu:
        a.batch = _fit_batch(a, nmb, max_seq, max(1, world))
    cfg = EngineConfig(model=a.model, pp_size=world, max_batch=a.batch, num_micro_batches=nmb, split_layers=not a.whole_layers,
                       max_seq_len=max_seq, prefill_chunk=max(2048, a.prompt_len), partition=a.partition,
                       use_graphs=not a.no_graph, device="cpu" if a.cpu else "auto",
                       backend="torch" if a.cpu else "auto", seed=0, dtype=a.dtype, decode_streams=a.decode_streams,
                       transport="gloo" if a.cpu else a.transport, kv_dtype="bf16" if a.cpu else a.kv_dtype)
    rank, world, rt, board, spec, cfg = attach_distributed(cfg)
    need = cfg.num_micro_batches * cfg.max_batch * ((cfg.max_seq_len + 31) // 32)
    have = getattr(rt.ex, "num_blocks", need)
    if have < need:                                  # the static block tables below index slot x blocks_per_seq
        raise SystemExit(f"bench.py: the KV pool holds {have} blocks, the schedule needs {need}; use a smaller --batch")
    if rank != 0:
        stage_main(rt)
        t = rt.barrier_times
        elapsed = (t[-1] - t[-2]) if len(t) >= 2 else 0.0
    else:
        from distributed_llm_inference_amd.parallel.shm import ST_READY
        board.set_state(0, ST_READY)
        drv = Driver(rt, cfg, a)
        drv.prefill()
        ctx = a.prompt_len
        for w in range(max(1, a.warmup)):           # the first round also uploads the batch composition
            drv.decode_round(first=(w == 0), ctx=ctx)
            ctx += 1
        drv.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            drv.decode_round(first=False, ctx=ctx, record=True)
            ctx += 1
        drv.barrier()
        elapsed = rt.barrier_times[-1] - rt.barrier_times[-2]
        lat = drv.latencies_ms()
        rt.execute(control(KIND_SHUTDOWN))
    # MAX over ranks
    if world > 1:
        import torch.distributed as dist
        dev = rt.device if (rt.on_gpu and dist.get_backend() == "nccl") else "cpu"
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        global_batch = cfg.num_micro_batches * cfg.max_batch
        tokens = global_batch * a.steps
        value = tokens / elapsed
        base = baseline_for(a.model)
        p50 = lat[len(lat) // 2] if lat else elapsed / a.steps * 1e3
        plans = bound = None
        try:
            from distributed_llm_inference_amd.models.partition import weak_scaling_bound
            from distributed_llm_inference_amd.parallel.launcher import plans_for
            pl = plans_for(cfg, spec)
            plans = [p.describe() for p in pl]
            bound = round(weak_scaling_bound(spec, pl, cfg.max_batch), 3)
        except Exception:
            pass
        print(json.dumps({
            "metric": f"output tokens/sec (whole node), {metric_model(spec.name)} layer-split PP=N",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (round(value / base, 1) if base else None),
            "vs_baseline_note": "baseline = the reference's own path re-enacted on the build host (CPU fp32, one request, "
                                "full recompute per token, JSON over HTTP; bench/baseline_numbers.json) — not a like-for-like "
                                "batch size", "dtype": "fp32" if a.cpu else {"bf16": "bf16", "fp8": "fp8 (e4m3 weights and activations, fp32 accumulate)", "fp8w": "fp8 weights, bf16 activations"}[a.dtype],
            "data": "synthetic prompts (random token ids), random-init weights",
            "p50_token_latency_ms": round(p50, 4),
            "config": {"model": spec.name, "global_batch": global_batch, "seq_len": a.prompt_len,
                       "parallelism": f"pp{world}", "micro_batches": cfg.num_micro_batches,
                       "micro_batch_size": cfg.max_batch, "layer_ranges": plans,
                       "weak_scaling_bound_from_partition": bound,
                       "sampling": "temperature 0.7, top-k 50, top-p 0.9", "hipgraph_decode": bool(cfg.use_graphs and not a.cpu), "decode_streams": cfg.decode_streams,
                       "backend": rt.ex.backend, "transport": getattr(rt.links, "transport", None) or cfg.resolved_transport(), "kv_cache": "fp32" if a.cpu else cfg.kv_dtype}}), flush=True)
    if world > 1:
        import torch.distributed as dist
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass
    rt.links.close()
    timer.cancel()


if __name__ == "__main__":
    main()
