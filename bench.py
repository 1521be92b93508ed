#!/usr/bin/env python3
"""Headline benchmark: output tokens/sec (whole node) and p50 per-token latency of Llama-3-8B
layer-split over N pipeline stages (PP = N GPUs of one node), bf16, synthetic prompts,
random-init weights.

    python bench.py --gpus 1 --steps 32 --warmup 4
    python bench.py --gpus N ...                    (no launcher: starts the N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --preflight            (devices, peer access, RCCL, every hop opened and timed; no model)

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

After the headline region the same pipeline, weights and KV cache run two short low-latency regions,
timed the same way and reported inside the same JSON line: `latency_point` — ONE micro-batch of 64
sequences in flight (what a lightly loaded server runs: decode steps sized to the rows in use) — and
`single_request` — one sequence alone, the reference's own operating point
(orchestration.py:109-196 serves one request at a time).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Hardware queues per process: a rank of a multi-rank run that has its GPU to itself gets at least 8 (its hop streams
# must not share a queue with its compute streams), everything else keeps what the environment says — the rule and the
# measurements behind it are in distributed_llm_inference_amd/__init__.py (ensure_hw_queues), applied there on import.
# RCCL and cross-process device buffers use dmabuf IPC handles; the legacy IPC mode is not supported by the driver here
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


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
    ap.add_argument("--latency-rows", type=int, default=64,
                    help="sequences of the single micro-batch of the low-latency region (0: skip both low-latency regions)")
    ap.add_argument("--latency-steps", type=int, default=24, help="timed steps of each low-latency region")
    ap.add_argument("--no-hop-probe", action="store_true", help="do not time the hops before the run")
    ap.add_argument("--preflight", action="store_true",
                    help="no model: report devices, peer access and RCCL, open + warm + time + close every hop, exit")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# Launching: `--gpus N` without a launcher starts the N ranks from here.
def self_launch(a) -> int:
    """Starts one fresh child process per rank (this same script, same arguments, the launcher's
    environment variables set) and waits for them. This process makes no GPU call, before or after:
    the children are created first thing, rank 0 inherits stdout (its JSON line is this command's
    output), the other ranks' stdout goes to stderr. Returns the exit code: 0 only if every rank
    returned 0; when one fails the others get a few seconds, then are ended by pid."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n = a.gpus
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_RUN_ID=f"self{os.getpid()}",
                   DLIA_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    deadline = time.monotonic() + a.max_seconds + 60.0
    failed_at = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        now = time.monotonic()
        if failed_at is None and (any(c not in (None, 0) for c in codes) or now > deadline):
            failed_at = now
            bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
            sys.stderr.write(f"bench.py: rank(s) {bad} failed (exit codes {codes}); ending the others\n" if bad
                             else "bench.py: ranks did not finish in time; ending them\n")
        if failed_at is not None and now - failed_at > 10.0:
            for p in procs:
                if p.poll() is None:
                    p.kill()                       # exactly the children started above
        time.sleep(0.05)
    codes = [p.returncode for p in procs]
    if any(codes):
        sys.stderr.write(f"bench.py: exit codes by rank: {codes}\n")
        return next(c for c in codes if c) if all(isinstance(c, int) for c in codes) else 1
    return 0


def metric_model(name: str) -> str:
    """'meta-llama/Meta-Llama-3-8B' -> 'Llama-3-8B' (the headline metric names the model)."""
    short = name.split("/")[-1]
    return short[5:] if short.startswith("Meta-") else short


def baseline_for(model: str):
    """Throughput (tok/s) of the reference's own CPU / JSON-over-HTTP path for this model, as measured
    by bench/reference_baseline.py and stored in bench/baseline_numbers.json; None if not measured."""
    path = os.path.join(ROOT, "bench", "baseline_numbers.json")
    try:
        with open(path) as f:
            return json.load(f).get(model, {}).get("tokens_per_sec")
    except Exception:
        return None


class Driver:
    """Rank 0: issues the static benchmark schedule through its own StageRuntime."""

    def __init__(self, rt, cfg, args):
        import torch
        from distributed_llm_inference_amd.runtime.batch import StepBatch
        self.torch, self.StepBatch = torch, StepBatch
        self.rt, self.cfg, self.a = rt, cfg, args
        self.links = rt.links
        self.B = cfg.max_batch
        self.nmb = cfg.num_micro_batches
        self.on_gpu = rt.on_gpu
        # (no stream that is not needed: streams share GPU_MAX_HW_QUEUES hardware queues, and one more than those made two
        # micro-batch streams share a queue — 9-13 % at 256-row steps)
        self.res_stream = torch.cuda.Stream(device=rt.device) if self.on_gpu and rt.links.num_stages > 1 else None
        self.fb = [None] * self.nmb
        self.step_id = 0
        self.lat_events = [[] for _ in range(self.nmb)]
        self.blocks_per_seq = (cfg.max_seq_len + 31) // 32

    def _probe(self, rows):
        return self.StepBatch(slots=list(range(rows)), q_lens=[1] * rows)       # "`rows` sequences decoding"

    def _desc(self, mb, rows, ctx, token_ids, replay=False):
        slots = [mb * self.B + i for i in range(rows)]
        self.step_id += 1
        return self.StepBatch(step_id=self.step_id, micro_batch=mb, slots=slots, q_lens=[1] * rows, ctx_lens=[ctx] * rows,
                              sample=[1] * rows, temperature=[0.7] * rows, top_k=[50] * rows, top_p=[0.9] * rows,
                              seeds=[1000 + s for s in slots], sample_steps=[0] * rows, token_ids=token_ids, replay=replay)

    def prefill(self):
        a, B, torch = self.a, self.B, self.torch
        g = torch.Generator().manual_seed(0)
        vocab = self.rt.ex.spec.vocab_size
        per_step = max(1, self.cfg.prefill_chunk // a.prompt_len)
        for mb in range(self.nmb):
            for s0 in range(0, B, per_step):
                n = min(per_step, B - s0)
                slots = [mb * B + s0 + i for i in range(n)]
                self.step_id += 1
                toks = torch.randint(10, vocab - 10, (n * a.prompt_len,), generator=g).tolist()
                d = self.StepBatch(step_id=self.step_id, micro_batch=mb, slots=slots, q_lens=[a.prompt_len] * n,
                                   ctx_lens=[0] * n, sample=[0] * n, temperature=[0.7] * n, top_k=[50] * n,
                                   top_p=[0.9] * n, seeds=slots, sample_steps=[0] * n, token_ids=toks,
                                   table_slots=slots,
                                   table_blocks=[list(range(s * self.blocks_per_seq, (s + 1) * self.blocks_per_seq))
                                                 for s in slots])
                self.rt.execute(d)

    def decode_round(self, first: bool, ctx: int, record: bool = False, mbs=None, rows: int = 0):
        """One decode step of each micro-batch in `mbs` (default: all) over its first `rows` rows
        (default: all). `first`: a new composition — per-row fields and explicit token ids travel
        with the descriptor; afterwards a step is "the same once more"."""
        torch = self.torch
        rows = rows or self.B
        for mb in (range(self.nmb) if mbs is None else mbs):
            if first:
                span = max(1, min(1000, self.rt.ex.spec.vocab_size - 20))
                d = self._desc(mb, rows, ctx, [(7 * (mb * self.B + i)) % span + 10 for i in range(rows)])
                fbw = None
            elif self.rt._graph_decode(self._probe(rows)):
                self.step_id += 1
                d = self.StepBatch.replay_of(self.step_id, mb, rows, rows)          # header-only descriptor: the stages replay their graphs
                fbw = self.fb[mb]
            else:
                d = self._desc(mb, rows, ctx, None, replay=True)
                fbw = self.fb[mb]
            with self.rt.stream_scope(d):
                self.rt.execute(d, fbw)
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
                    self.fb[mb] = _EvHandle(got, torch)
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
        from distributed_llm_inference_amd.runtime.batch import KIND_BARRIER, control
        self.rt.execute(control(KIND_BARRIER))

    def latencies_ms(self):
        out = []
        for evs in self.lat_events:
            for a, b in zip(evs, evs[1:]):
                out.append(a.elapsed_time(b) if hasattr(a, "elapsed_time") else (b - a) * 1e3)
        self.lat_events = [[] for _ in range(self.nmb)]
        return sorted(out)

    def region(self, ctx: int, steps: int, warm: int, mbs=None, rows: int = 0, new_composition: bool = True):
        """`warm` untimed steps (the first uploads the composition), barrier, `steps` timed steps,
        barrier. Returns (context length afterwards, seconds between the barriers, step latencies)."""
        for w in range(max(1, warm)):
            self.decode_round(first=(w == 0 and new_composition), ctx=ctx, mbs=mbs, rows=rows)
            ctx += 1
        self.barrier()
        for _ in range(steps):
            self.decode_round(first=False, ctx=ctx, record=True, mbs=mbs, rows=rows)
            ctx += 1
        self.barrier()
        return ctx, self.rt.barrier_times[-1] - self.rt.barrier_times[-2], self.latencies_ms()


class _EvHandle:
    def __init__(self, ev, torch):
        self.ev, self.torch = ev, torch

    def wait(self):
        self.torch.cuda.current_stream().wait_event(self.ev)

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


def _fit_batch(a, nmb: int, max_seq: int, world: int, quiet: bool = False) -> int:
    """The KV cache holds every in-flight sequence for prompt + warm-up + timed steps. A long run
    (--steps in the hundreds) at 2 x 1024 sequences per GPU, or a 70B model on one GPU, would ask
    for more than the device has: halve the micro-batch until the largest stage's share fits in 85 %
    of what its weights leave, and say so (the size used is reported in the JSON line's config)."""
    import torch
    from distributed_llm_inference_amd.config import EngineConfig
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
    if batch != a.batch and not quiet and int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write(f"bench.py: KV cache for {nmb} x {a.batch} sequences of {max_seq} tokens does not fit; "
                         f"using micro-batches of {batch}\n")
    return batch


def _fit_schedule(a, nmb: int, max_seq: int, world: int, spares: int):
    """(micro-batch size, micro-batches) that fit. When a long run's KV cache does not hold every micro-batch
    at full size the spare micro-batches go first (two of them are worth ~3 % of a full pipeline,
    bench/pipeline_sim.py); the micro-batch is halved (2048 -> 1024 rows: ~5 %) only if that is not enough."""
    want = a.batch
    if spares and _fit_batch(a, nmb, max_seq, world, quiet=True) < want and \
            _fit_batch(a, nmb - spares, max_seq, world, quiet=True) == want:
        nmb -= spares
        if int(os.environ.get("RANK", "0")) == 0:
            sys.stderr.write(f"bench.py: KV cache for {nmb + spares} x {want} sequences of {max_seq} tokens does not fit; "
                             f"running {nmb} micro-batches (no spares)\n")
    return _fit_batch(a, nmb, max_seq, world), nmb


def _combined_hops(ranks):
    from distributed_llm_inference_amd.parallel.preflight import combine_hops
    return combine_hops([r["hops_measured"] for r in ranks])


def _say(rank: int, msg: str):
    sys.stderr.write(f"[bench rank {rank}] {msg}\n")
    sys.stderr.flush()


def preflight(a, world: int, emit=print) -> int:
    """No model: what this rank sees (devices, peer access, RCCL), then — with more than one rank —
    the default group, every hop opened, warmed, timed and closed. Everything goes to stderr, rank 0
    also prints one JSON line with the collected report."""
    import torch
    from distributed_llm_inference_amd.parallel.preflight import combine_hops, device_report, probe_hops
    rank = int(os.environ.get("RANK", "0"))
    t0 = time.perf_counter()
    rep = {"rank": rank, **device_report(peer_matrix=(rank == 0) and not a.cpu)} if not a.cpu else {"rank": rank, "device_count": 0}
    if rank == 0:
        _say(0, f"devices: {rep.get('device_count')}  rccl: {rep.get('rccl_version')}")
        for d in rep.get("devices", []):
            _say(0, f"  cuda:{d['index']} {d['name']} {d['gcn_arch']} {d['memory_gib']} GiB, {d['cus']} CUs")
        if rep.get("peer_access"):
            for i, row in enumerate(rep["peer_access"]):
                _say(0, f"  peer access from {i}: " + " ".join("1" if x else "0" for x in row))
    hops, transport = [], None
    if world > 1:
        from distributed_llm_inference_amd.config import EngineConfig
        from distributed_llm_inference_amd.parallel import launcher as L
        cfg = EngineConfig(model=a.model, pp_size=world, device="cpu" if a.cpu else "auto", transport="gloo" if a.cpu else a.transport)
        links, device = L.open_links_only(cfg, rank, world)
        transport = getattr(links, "transport", None) or cfg.resolved_transport()
        spec = L.resolve_spec(cfg)
        batch = a.batch or (8 if a.cpu else 2048)
        raw = probe_hops(links, device, big_bytes=2 * spec.hidden_size * batch)
        import torch.distributed as dist
        allh = [None] * world
        dist.all_gather_object(allh, {"rank": rank, "hops": raw, "transport": transport})
        links.close()
        dist.barrier()
        dist.destroy_process_group()
        hops = combine_hops([r["hops"] for r in allh])
        if rank == 0:
            _say(0, f"transport by rank: {[r['transport'] for r in allh]}")
            for h in hops:
                _say(0, f"hop {h['hop']}: {h['bytes']} bytes one way in {h['us_min']} us (median {h['us_p50']}; {h['GBps']} GB/s)")
    if rank == 0:
        emit(json.dumps({"preflight": rep, "n_gpus": world, "transport": transport, "hops": hops,
                         "seconds": round(time.perf_counter() - t0, 2)}))
    return 0


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        # no launcher: this process becomes one (it never touches a GPU; nothing below runs in it)
        raise SystemExit(self_launch(a))
    world = int(env_world or "1")
    if world != a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to run "
                         f"a different job than the one asked for\n")
        raise SystemExit(2)
    timer = _deadline(a.max_seconds)
    # stdout carries the one JSON line and nothing else: whatever libraries print there while the
    # pipeline comes up (gloo reports its connections on stdout) goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line: str):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(line, flush=True)
        os.dup2(2, 1)

    if a.preflight:
        rc = preflight(a, world, emit)
        timer.cancel()
        raise SystemExit(rc)

    import torch
    from distributed_llm_inference_amd.config import EngineConfig
    from distributed_llm_inference_amd.parallel.launcher import attach_distributed
    from distributed_llm_inference_amd.parallel.pipeline import stage_main
    from distributed_llm_inference_amd.runtime.batch import KIND_SHUTDOWN, control

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
    lat_rows = min(a.latency_rows, a.batch) if a.latency_rows > 0 else 0
    lat_warm = 3
    extra = 2 * (a.latency_steps + lat_warm) if lat_rows else 0
    max_seq = ((a.prompt_len + a.steps + a.warmup + extra + 2 + 31) // 32) * 32
    if not a.cpu:
        spares = 2 if (world > 1 and wide and not a.micro_batches) else 0
        a.batch, nmb = _fit_schedule(a, nmb, max_seq, max(1, world), spares)
        lat_rows = min(lat_rows, a.batch)
    # the low-latency regions run decode steps of `lat_rows` rows and of one row on micro-batch 0:
    # captured at first use (only micro-batch 0 ever runs them), like the sizes a server keeps for light load
    buckets = sorted({b for b in (1, lat_rows) if 0 < b < a.batch}) if lat_rows else []
    cfg = EngineConfig(model=a.model, pp_size=world, max_batch=a.batch, num_micro_batches=nmb, split_layers=not a.whole_layers,
                       max_seq_len=max_seq, prefill_chunk=max(2048, a.prompt_len), partition=a.partition,
                       use_graphs=not a.no_graph, device="cpu" if a.cpu else "auto",
                       backend="torch" if a.cpu else "auto", seed=0, dtype=a.dtype, decode_streams=a.decode_streams,
                       transport="gloo" if a.cpu else a.transport, kv_dtype="bf16" if a.cpu else a.kv_dtype,
                       decode_buckets=buckets, warm_decode_graphs=False)
    t_up = time.perf_counter()
    rank, world, rt, board, spec, cfg = attach_distributed(cfg)
    need = cfg.num_micro_batches * cfg.max_batch * ((cfg.max_seq_len + 31) // 32)
    have = getattr(rt.ex, "num_blocks", need)
    if have < need:                                  # the static block tables below index slot x blocks_per_seq
        raise SystemExit(f"bench.py: the KV pool holds {have} blocks, the schedule needs {need}; use a smaller --batch")
    transport = getattr(rt.links, "transport", None) or cfg.resolved_transport()
    hops = []
    if world > 1 and not a.no_hop_probe:
        # one hidden-state message and one 8 KB message per hop, after the hops' own warm-up: what a
        # stage-to-stage hand-off costs on this machine, measured before anything else uses the links
        from distributed_llm_inference_amd.parallel.preflight import probe_hops
        hops = probe_hops(rt.links, rt.device if rt.on_gpu else None, big_bytes=2 * spec.hidden_size * cfg.max_batch)
        if rank == 0:
            _say(0, f"{world} ranks up on the {transport} transport")
    bringup_s = time.perf_counter() - t_up
    regions = {}
    if rank != 0:
        stage_main(rt)
        t = rt.barrier_times
        for i, name in enumerate(("headline", "latency_point", "single_request")):
            if len(t) >= 2 * i + 2:
                regions[name] = t[2 * i + 1] - t[2 * i]
        lat = {}
    else:
        from distributed_llm_inference_amd.parallel.shm import ST_READY
        board.set_state(0, ST_READY)
        drv = Driver(rt, cfg, a)
        drv.prefill()
        lat = {}
        ctx, regions["headline"], lat["headline"] = drv.region(a.prompt_len, a.steps, a.warmup)
        if lat_rows:
            ctx, regions["latency_point"], lat["latency_point"] = drv.region(ctx, a.latency_steps, lat_warm, mbs=[0], rows=lat_rows)
            ctx, regions["single_request"], lat["single_request"] = drv.region(ctx, a.latency_steps, lat_warm, mbs=[0], rows=1)
        rt.execute(control(KIND_SHUTDOWN))
    # MAX over ranks of every region; per-rank facts for the report
    mine = {"rank": rank, "device": str(rt.device), "transport": transport, "kv_blocks": int(getattr(rt.ex, "num_blocks", 0)),
            "seconds": {k: round(v, 6) for k, v in regions.items()}, "hops_measured": hops,
            "graph_replays": int(getattr(rt.ex, "graph_replays", 0)),
            "native_loop_steps": int(getattr(rt, "native_stats", {}).get("native_steps", 0)) if hasattr(rt, "native_stats") else 0,
            "bringup_s": round(bringup_s, 2)}
    ranks = [mine]
    if world > 1:
        import torch.distributed as dist
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        for name in list(regions):
            regions[name] = max(r["seconds"].get(name, 0.0) for r in ranks)
    if rank == 0:
        elapsed = regions["headline"]
        global_batch = cfg.num_micro_batches * cfg.max_batch
        tokens = global_batch * a.steps
        value = tokens / elapsed
        base = baseline_for(a.model)
        hl = lat.get("headline") or []
        p50 = hl[len(hl) // 2] if hl else elapsed / a.steps * 1e3
        plans = bound = None
        try:
            from distributed_llm_inference_amd.models.partition import weak_scaling_bound
            from distributed_llm_inference_amd.parallel.launcher import plans_for
            pl = plans_for(cfg, spec)
            plans = [p.describe() for p in pl]
            bound = round(weak_scaling_bound(spec, pl, cfg.max_batch), 3)
        except Exception:
            pass

        def point(name, rows):
            if name not in regions:
                return None
            ms = regions[name] / a.latency_steps * 1e3
            l = lat.get(name) or []
            return {"batch": rows, "micro_batches": 1, "steps": a.latency_steps, "ms_per_token": round(ms, 4),
                    "p50_token_latency_ms": round(l[len(l) // 2], 4) if l else round(ms, 4),
                    "tok_s": round(rows / (ms * 1e-3), 2)}

        single = point("single_request", 1)
        out = {
            "metric": f"output tokens/sec (whole node), {metric_model(spec.name)} layer-split PP=N",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (round(value / base, 1) if base else None),
            "vs_baseline_note": "baseline = the reference's own path re-enacted on the build host (CPU fp32, ONE request, "
                                "full recompute per token, JSON over HTTP; bench/baseline_numbers.json); `value` is the "
                                "throughput of every in-flight sequence together, the like-for-like figure is "
                                "single_request.vs_baseline (one sequence alone on the same pipeline)",
            "dtype": "fp32" if a.cpu else {"bf16": "bf16", "fp8": "fp8 (e4m3 weights and activations, fp32 accumulate)", "fp8w": "fp8 weights, bf16 activations"}[a.dtype],
            "data": "synthetic prompts (random token ids), random-init weights",
            "p50_token_latency_ms": round(p50, 4),
            "latency_point": point("latency_point", lat_rows),
            "single_request": ({**single, "vs_baseline": (round(single["tok_s"] / base, 1) if base else None)} if single else None),
            "config": {"model": spec.name, "global_batch": global_batch, "seq_len": a.prompt_len,
                       "parallelism": f"pp{world}", "micro_batches": cfg.num_micro_batches,
                       "micro_batch_size": cfg.max_batch, "layer_ranges": plans,
                       "weak_scaling_bound_from_partition": bound,
                       "sampling": "temperature 0.7, top-k 50, top-p 0.9", "hipgraph_decode": bool(cfg.use_graphs and not a.cpu), "decode_streams": cfg.decode_streams,
                       "backend": rt.ex.backend, "transport": transport, "kv_cache": "fp32" if a.cpu else cfg.kv_dtype,
                       "kv_blocks_per_stage": mine["kv_blocks"],
                       "launcher": "self" if os.environ.get("DLIA_BENCH_SELF_LAUNCHED") == "1" else ("torch.distributed.run" if env_world is not None else "none")},
            # what a reader needs to check a scaling run: every rank that took part, the transport each ended up
            # on (fallbacks included), what a message cost on each hop, and each stage's own clock over the regions
            "ranks": [{"rank": r["rank"], "device": r["device"], "transport": r["transport"], "kv_blocks": r["kv_blocks"],
                       "ms_per_step": round(r["seconds"].get("headline", 0.0) / a.steps * 1e3, 4),
                       "graph_replays": r["graph_replays"], "native_loop_steps": r["native_loop_steps"],
                       "bringup_s": r["bringup_s"]} for r in ranks],
            "hops": _combined_hops(ranks),
        }
        emit(json.dumps(out))
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
