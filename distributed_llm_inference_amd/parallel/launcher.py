"""Bringing a pipeline up: executors, links, stage processes, the engine.

Three ways to run P stages:

* `LocalPipeline`  — one process; stage 0 is driven by the engine, stages 1..P-1 are threads over
  loopback links. For tests and for P = 1 serving.
* `spawn_pipeline` — one OS process per stage (rank = stage = GPU), started here with
  multiprocessing; `torch.distributed` is initialised over 127.0.0.1 (backend nccl = RCCL over
  xGMI on GPUs, gloo on CPU); control descriptors go through shared-memory rings.
* `attach_distributed` — the same, for processes some launcher (torchrun) already started: reads
  RANK / WORLD_SIZE / MASTER_* from the environment.

The reference's equivalent is three Colab notebooks started by hand, each loading the full model,
exposing Flask through ngrok, with worker URLs pasted into the orchestrator source
(orchestration.py:359-372; Worker1.py:248-266; Worker2.py:241-273).
"""
from __future__ import annotations

import dataclasses
import os
import socket
import threading
import time
import uuid
from typing import List, Optional, Tuple

import torch

from ..config import EngineConfig
from ..models.partition import StagePlan, make_plans
from ..models.spec import ModelSpec
from ..models.weights import StageWeights, load_stage_weights, random_stage_weights
from ..runtime.tokenizer import load_tokenizer
from .pipeline import StageRuntime, stage_main
from .shm import ST_ERROR, ST_READY, StatusBoard
from ..utils.tracing import make_tracer
from .transport import PipelineLinks, make_dist_links, make_loopback_links


def has_checkpoint(path: str) -> bool:
    if not os.path.isdir(path):
        return False
    return any(f.endswith(".safetensors") or f == "pytorch_model.bin" for f in os.listdir(path))


def resolve_spec(cfg: EngineConfig) -> ModelSpec:
    return ModelSpec.from_pretrained(cfg.model)


def plans_for(cfg: EngineConfig, spec: ModelSpec) -> List[StagePlan]:
    return make_plans(spec, cfg.pp_size, mode=cfg.partition, ranges=cfg.layer_ranges)


def kv_blocks_for(cfg: EngineConfig) -> int:
    """Blocks every stage's cache is built with. The scheduler's allocator uses the same number,
    so a block id valid on one stage is valid on all."""
    if cfg.kv_blocks > 0:
        return cfg.kv_blocks
    per_seq = (cfg.max_seq_len + 31) // 32
    return cfg.num_micro_batches * cfg.max_batch * per_seq


def load_weights(cfg: EngineConfig, spec: ModelSpec, plan: StagePlan, device: str, dtype) -> StageWeights:
    use_ckpt = cfg.weights == "checkpoint" or (cfg.weights == "auto" and has_checkpoint(cfg.model))
    if use_ckpt:
        return load_stage_weights(cfg.model, spec, plan, device="cpu", dtype=dtype)
    return random_stage_weights(spec, plan, seed=cfg.seed, device=device, dtype=dtype)


def build_executor(cfg: EngineConfig, spec: ModelSpec, plan: StagePlan, device: Optional[str] = None):
    backend = cfg.resolved_backend()
    dev = device or cfg.resolved_device()
    if backend == "hip":
        from ..models.stage_gpu import GpuStage
        w = load_weights(cfg, spec, plan, dev, torch.bfloat16)
        return GpuStage(spec, plan, w, device=dev, max_slots=cfg.num_micro_batches * cfg.max_batch,
                        max_seq_len=cfg.max_seq_len, num_blocks=kv_blocks_for(cfg),
                        max_tokens=cfg.prefill_chunk, max_batch=cfg.max_batch, use_graphs=cfg.use_graphs,
                        kv_mem_fraction=cfg.kv_mem_fraction, weight_dtype=cfg.dtype if cfg.dtype in ("fp8", "fp8w") else "bf16",
                        kv_dtype=cfg.kv_dtype)
    if cfg.kv_dtype != "bf16":
        raise ValueError(f"kv_dtype {cfg.kv_dtype!r} needs the hip backend (the torch executor keeps its cache in the compute dtype)")
    from ..models.stage_ref import RefStage
    dt = torch.float32 if cfg.dtype in ("fp32", "bf16") and dev == "cpu" else torch.bfloat16
    w = load_weights(cfg, spec, plan, "cpu", torch.float32)
    return RefStage(spec, plan, w, dtype=dt, device=dev, max_seq_len=cfg.max_seq_len)


def make_tokenizer(cfg: EngineConfig, spec: ModelSpec):
    path = cfg.tokenizer or (cfg.model if os.path.isdir(cfg.model) else None)
    return load_tokenizer(path, spec.vocab_size, spec.eos_token_id, spec.bos_token_id)


# ---------------------------------------------------------------------------------------------
class LocalPipeline:
    """All stages in this process (stages > 0 on threads, loopback links)."""

    def __init__(self, cfg: EngineConfig, executors: Optional[list] = None):
        from ..runtime.engine import Engine
        self.cfg = cfg
        self.spec = resolve_spec(cfg)
        self.plans = plans_for(cfg, self.spec)
        self.job = f"dlia_{os.getpid()}_{uuid.uuid4().hex[:8]}"
        self.board = StatusBoard(self.job + "_b", create=True, num_stages=cfg.pp_size)
        links = make_loopback_links(cfg.pp_size)
        exs = executors or [build_executor(cfg, self.spec, p) for p in self.plans]
        self.executors = exs
        self.tracer = make_tracer(cfg.trace_path)      # one process: all stages share the tracer
        self.runtimes = [StageRuntime(ex, l, board=self.board, fault=cfg.fault_inject, tracer=self.tracer, decode_streams=cfg.decode_streams, debug_top5=cfg.debug_top5,
                                      trace_path=cfg.trace_path if l.stage == 0 else None)
                         for ex, l in zip(exs, links)]
        for p in self.plans:
            self.board.register(p.stage, p.layer_start, p.layer_end)
        self.board.set_state(0, ST_READY)
        self.threads = []
        for rt in self.runtimes[1:]:
            t = threading.Thread(target=self._run_stage, args=(rt,), name=f"dlia-stage{rt.stage}", daemon=True)
            t.start()
            self.threads.append(t)
        self.tokenizer = make_tokenizer(cfg, self.spec)
        self.engine = Engine(cfg, self.runtimes[0], self.tokenizer, kv_blocks_for(cfg), board=self.board,
                             tracer=self.tracer).start()

    @staticmethod
    def _run_stage(rt):
        try:
            stage_main(rt)
        except BaseException:
            pass

    def close(self):
        self.engine.stop()
        for t in self.threads:
            t.join(5.0)
        self.runtimes[0].dump_trace()
        self.board.close()


# ---------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_distributed(rank: int, world: int, backend: str, master_port: int, device: Optional[str]):
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(master_port)       # not setdefault: a rebuilt pipeline gets a new port
    kw = {}
    if backend == "nccl" and device is not None:
        torch.cuda.set_device(torch.device(device))
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=600), **kw)
    return dist


def stage_process_setup(cfg: EngineConfig, rank: int, world: int, job: str, master_port: int,
                        local_rank: Optional[int] = None):
    """Common bring-up of one stage process. Returns (spec, plan, executor, links, board)."""
    dev_kind = cfg.resolved_device()
    transport = cfg.resolved_transport()
    backend = {"rccl": "nccl", "nccl": "nccl", "gloo": "gloo"}.get(transport, "gloo")
    device = "cpu"
    if dev_kind == "cuda":
        idx = rank if local_rank is None else local_rank
        ndev = torch.cuda.device_count()
        if idx >= ndev > 0 and (backend == "gloo" or os.environ.get("DLIA_SHARE_GPU") == "1"):
            # more stages than GPUs: only the host-staged transport can share a device between ranks
            # (RCCL refuses two ranks on one GPU); used to rehearse a multi-stage job on a small box
            idx %= ndev
        device = f"cuda:{idx}"
    if dev_kind == "cuda":
        torch.cuda.set_device(torch.device(device))
    spec = resolve_spec(cfg)
    plan = plans_for(cfg, spec)[rank]
    board = StatusBoard(job + "_b", create=(rank == 0), num_stages=world)
    board.register(rank, plan.layer_start, plan.layer_end)
    try:
        if world > 1:
            init_distributed(rank, world, backend, master_port, device if dev_kind == "cuda" else None)
        ex = build_executor(cfg, spec, plan, device=device)
        links = (make_dist_links(rank, world, backend, job, host_staged=(dev_kind == "cuda" and backend == "gloo"))
                 if world > 1 else make_loopback_links(1)[0])
    except BaseException as e:
        board.set_state(rank, ST_ERROR, f"{type(e).__name__}: {e}")
        raise
    return spec, plan, ex, links, board


def _worker_entry(cfg_dict: dict, rank: int, world: int, job: str, master_port: int):
    cfg = EngineConfig.from_dict(cfg_dict)
    spec, plan, ex, links, board = stage_process_setup(cfg, rank, world, job, master_port)
    rt = StageRuntime(ex, links, board=board, fault=cfg.fault_inject, tracer=make_tracer(cfg.trace_path), decode_streams=cfg.decode_streams, debug_top5=cfg.debug_top5,
                      trace_path=cfg.trace_path)
    try:
        stage_main(rt)
    finally:
        links.close()


class SpawnedPipeline:
    """Rank 0 (this process: scheduler + stage 0) plus P-1 spawned stage processes."""

    def __init__(self, cfg: EngineConfig):
        import multiprocessing as mp
        from ..runtime.engine import Engine
        self.cfg = cfg
        world = cfg.pp_size
        self.job = f"dlia_{os.getpid()}_{uuid.uuid4().hex[:8]}"
        port = _free_port()
        ctx = mp.get_context("spawn")
        self.procs = []
        # the status board must exist before the children attach
        self.board = StatusBoard(self.job + "_b", create=True, num_stages=world)
        for r in range(1, world):
            p = ctx.Process(target=_worker_entry, args=(cfg.to_dict(), r, world, self.job, port), daemon=True)
            p.start()
            self.procs.append(p)
        dev_kind = cfg.resolved_device()
        device = "cuda:0" if dev_kind == "cuda" else "cpu"
        backend = {"rccl": "nccl", "nccl": "nccl", "gloo": "gloo"}.get(cfg.resolved_transport(), "gloo")
        self.spec = resolve_spec(cfg)
        self.plan = plans_for(cfg, self.spec)[0]
        self.board.register(0, self.plan.layer_start, self.plan.layer_end)
        try:
            init_distributed(0, world, backend, port, device if dev_kind == "cuda" else None)
            ex = build_executor(cfg, self.spec, self.plan, device=device)
            self.links = make_dist_links(0, world, backend, self.job,
                                         host_staged=(dev_kind == "cuda" and backend == "gloo"))
        except BaseException as e:
            self.board.set_state(0, ST_ERROR, f"{type(e).__name__}: {e}")
            raise
        self.tracer = make_tracer(cfg.trace_path)
        self.runtime = StageRuntime(ex, self.links, board=self.board, fault=cfg.fault_inject, tracer=self.tracer, decode_streams=cfg.decode_streams, debug_top5=cfg.debug_top5,
                                    trace_path=cfg.trace_path)
        self.board.set_state(0, ST_READY)
        self.tokenizer = make_tokenizer(cfg, self.spec)
        self.engine = Engine(cfg, self.runtime, self.tokenizer, kv_blocks_for(cfg), board=self.board,
                             tracer=self.tracer).start()

    def wait_ready(self, timeout: float = 300.0):
        t0 = time.time()
        while time.time() - t0 < timeout:
            states = [self.board.read(s)["state"] for s in range(self.cfg.pp_size)]
            if all(s == "ready" for s in states):
                return True
            if any(s == "error" for s in states):
                raise RuntimeError(f"stage bring-up failed: {[self.board.read(s) for s in range(self.cfg.pp_size)]}")
            time.sleep(0.05)
        raise TimeoutError("pipeline stages did not come up")

    def close(self):
        self.engine.stop()
        for p in self.procs:
            p.join(10.0)
            if p.is_alive():
                p.terminate()
        self.runtime.dump_trace()                  # after the joins: folds in the other stages' files
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass
        self.links.close()
        self.board.close()


def attach_distributed(cfg: EngineConfig):
    """For ranks started by an external launcher. Returns (rank, world, runtime, board, spec):
    rank 0 builds an Engine on its runtime; other ranks call `stage_main(runtime)`."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    port = int(os.environ.get("MASTER_PORT", "29500"))
    job = f"dlia_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}_{os.getppid()}"
    if cfg.pp_size != world:
        cfg = dataclasses.replace(cfg, pp_size=world)
    # rank 0 creates the status board; the other ranks attach to it with retries (StatusBoard)
    spec, plan, ex, links, board = stage_process_setup(cfg, rank, world, job, port, local_rank=local_rank)
    rt = StageRuntime(ex, links, board=board, fault=cfg.fault_inject, tracer=make_tracer(cfg.trace_path), decode_streams=cfg.decode_streams, debug_top5=cfg.debug_top5,
                      trace_path=cfg.trace_path)
    return rank, world, rt, board, spec, cfg


def make_pipeline(cfg: EngineConfig):
    """The right pipeline object for `cfg`: in-process for one stage or loopback transport,
    spawned processes otherwise."""
    if cfg.pp_size == 1 or cfg.resolved_transport() == "loopback":
        return LocalPipeline(cfg)
    p = SpawnedPipeline(cfg)
    p.wait_ready()
    return p
