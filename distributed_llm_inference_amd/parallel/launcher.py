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
import gc
import os
import socket
import threading
import time
import uuid
from typing import List, Optional

import torch

from ..config import EngineConfig
from ..models.partition import StagePlan, make_plans
from ..models.spec import ModelSpec
from ..models.weights import StageWeights, load_stage_weights, random_stage_weights
from ..runtime.tokenizer import load_tokenizer
from .pipeline import StageRuntime, stage_main
from .shm import ST_ERROR, ST_READY, StatusBoard
from ..utils.tracing import make_tracer
from .transport import make_dist_links, make_loopback_links


def has_checkpoint(path: str) -> bool:
    if not os.path.isdir(path):
        return False
    return any(f.endswith(".safetensors") or f == "pytorch_model.bin" for f in os.listdir(path))


def resolve_spec(cfg: EngineConfig) -> ModelSpec:
    return ModelSpec.from_pretrained(cfg.model)


def plans_for(cfg: EngineConfig, spec: ModelSpec) -> List[StagePlan]:
    return make_plans(spec, cfg.pp_size, mode=cfg.partition, ranges=cfg.layer_ranges, split_layers=cfg.split_layers,
                      rows=cfg.max_batch)


def kv_blocks_for(cfg: EngineConfig) -> int:
    """KV blocks when the count does not come from device memory: the configured number, or (torch
    executor, whose cache is not paged) enough for every slot at full length."""
    if cfg.kv_blocks > 0:
        return cfg.kv_blocks
    per_seq = (cfg.max_seq_len + 31) // 32
    return cfg.num_micro_batches * cfg.max_batch * per_seq


def ctrl_slot_bytes(cfg: EngineConfig) -> int:
    """Slot size of the descriptor rings: the largest step descriptor this configuration can emit
    (per-row fields of a full micro-batch, a prefill chunk's token ids, the block table of the
    longest admissible prompt for every sequence of a chunk, growth entries), rounded up to a
    power of two. Checked here, at configuration time, not when a descriptor first overflows."""
    from ..runtime.engine import Engine
    per_seq_blocks = (cfg.max_seq_len + 31) // 32
    chunk = max(cfg.prefill_chunk, cfg.max_batch)
    n = 64 + 40 * cfg.max_batch + 4 * chunk + 4 * (2 * cfg.max_batch + chunk // 32 + per_seq_blocks + cfg.max_batch) \
        + 12 * cfg.max_batch + 4096 \
        + 4 * Engine.MOVE_BUDGET + 8 * min(cfg.max_batch, Engine.MOVE_BUDGET)      # tables of rows moved at a recomposition
    size = 1 << 16
    while size < n:
        size <<= 1
    return size


def load_weights(cfg: EngineConfig, spec: ModelSpec, plan: StagePlan, device: str, dtype) -> StageWeights:
    use_ckpt = cfg.weights == "checkpoint" or (cfg.weights == "auto" and has_checkpoint(cfg.model))
    if use_ckpt:
        return load_stage_weights(cfg.model, spec, plan, device="cpu", dtype=dtype)
    return random_stage_weights(spec, plan, seed=cfg.seed, device=device, dtype=dtype)


def build_executor(cfg: EngineConfig, spec: ModelSpec, plan: StagePlan, device: Optional[str] = None,
                   defer_kv: bool = False):
    """The stage executor for `plan`. HIP backend: `cfg.kv_blocks` > 0 fixes the cache size, else it
    is sized from the device memory left after weights and workspaces (`kv_mem_fraction`); with
    `defer_kv` the cache is allocated later by `settle_kv` (the stages first agree on one count)."""
    backend = cfg.resolved_backend()
    dev = device or cfg.resolved_device()
    if backend == "hip":
        from ..models.stage_gpu import GpuStage
        w = load_weights(cfg, spec, plan, dev, torch.bfloat16)
        return GpuStage(spec, plan, w, device=dev, max_slots=cfg.num_micro_batches * cfg.max_batch,
                        max_seq_len=cfg.max_seq_len, num_blocks=cfg.kv_blocks if cfg.kv_blocks > 0 else None,
                        max_tokens=cfg.prefill_chunk, max_batch=cfg.max_batch, use_graphs=cfg.use_graphs,
                        kv_mem_fraction=cfg.kv_mem_fraction, weight_dtype=cfg.dtype if cfg.dtype in ("fp8", "fp8w") else "bf16",
                        kv_dtype=cfg.kv_dtype, decode_states=cfg.num_micro_batches, defer_kv=defer_kv,
                        graph_sizes=cfg.graph_sizes())
    if cfg.kv_dtype != "bf16":
        raise ValueError(f"kv_dtype {cfg.kv_dtype!r} needs the hip backend (the torch executor keeps its cache in the compute dtype)")
    from ..models.stage_ref import RefStage
    dt = torch.float32 if cfg.dtype in ("fp32", "bf16") and dev == "cpu" else torch.bfloat16
    w = load_weights(cfg, spec, plan, "cpu", torch.float32)
    return RefStage(spec, plan, w, dtype=dt, device=dev, max_seq_len=cfg.max_seq_len)


def kv_budget(cfg: EngineConfig, ex, share: float = 1.0) -> int:
    """Blocks this stage could hold: from its device memory (HIP) or the configured / full-length count."""
    if cfg.kv_blocks > 0 or not hasattr(ex, "kv_budget_blocks"):
        return kv_blocks_for(cfg)
    return ex.kv_budget_blocks(share)


def settle_kv(ex, blocks: int, warm: bool = True):
    if hasattr(ex, "allocate_kv") and ex.kv is None:
        ex.allocate_kv(blocks)
    if warm and hasattr(ex, "warm_decode_states"):
        ex.warm_decode_states()          # with decode buckets: every size's state and graph exist before the first request


def make_tokenizer(cfg: EngineConfig, spec: ModelSpec):
    path = cfg.tokenizer or (cfg.model if os.path.isdir(cfg.model) else None)
    return load_tokenizer(path, spec.vocab_size, spec.eos_token_id, spec.bos_token_id)


# ---------------------------------------------------------------------------------------------
class LocalPipeline:
    """All stages in this process (stages > 0 on threads, loopback links)."""

    def __init__(self, cfg: EngineConfig, executors: Optional[list] = None):
        from ..runtime.engine import Engine
        self.cfg = cfg
        # What an earlier pipeline of this process left to the cyclic collector (executors, graphs, device
        # buffers caught in a reference cycle) is released now, on this thread, not at some point during this
        # pipeline's run on whichever of its threads trips the collector.
        gc.collect()
        self.spec = resolve_spec(cfg)
        self.plans = plans_for(cfg, self.spec)
        self.job = f"dlia_{os.getpid()}_{uuid.uuid4().hex[:8]}"
        self.board = StatusBoard(self.job + "_b", create=True, num_stages=cfg.pp_size)
        exs = executors or [build_executor(cfg, self.spec, p, defer_kv=True) for p in self.plans]
        dev = getattr(exs[0], "device", None)
        on_gpu = dev is not None and torch.device(dev).type == "cuda" and hasattr(exs[0], "decode_step")
        act_bytes = 2 * self.spec.hidden_size * max(cfg.prefill_chunk, cfg.max_batch)
        links = make_loopback_links(cfg.pp_size, device=dev if on_gpu else None, max_bytes=act_bytes,
                                    fb_bytes=4 * max(cfg.max_batch, 64), job=self.job, ctrl_slot_bytes=ctrl_slot_bytes(cfg),
                                    timeout_s=cfg.hop_timeout_s * cfg.pp_size)
        # every stage's cache holds the same number of blocks: the smallest budget (the stages of an
        # in-process pipeline split one device between them)
        self.kv_blocks = min(kv_budget(cfg, ex, 1.0 / len(exs)) for ex in exs)
        for ex in exs:
            settle_kv(ex, self.kv_blocks, cfg.warm_decode_graphs)
        self.executors = exs
        self._own_executors = executors is None
        self.tracer = make_tracer(cfg.trace_path)      # one process: all stages share the tracer
        self.runtimes = [StageRuntime(ex, l, board=self.board, fault=cfg.fault_inject, tracer=self.tracer, decode_streams=cfg.decode_streams, debug_top5=cfg.debug_top5,
                                      trace_path=cfg.trace_path if l.stage == 0 else None)
                         for ex, l in zip(exs, links)]
        for p in self.plans:
            self.board.register(p.stage, p.layer_start, p.layer_end)
            self.board.set_kv_blocks(p.stage, self.kv_blocks)
        self.board.set_state(0, ST_READY)
        self.threads = []
        for rt in self.runtimes[1:]:
            t = threading.Thread(target=self._run_stage, args=(rt,), name=f"dlia-stage{rt.stage}", daemon=True)
            t.start()
            self.threads.append(t)
        self.tokenizer = make_tokenizer(cfg, self.spec)
        self.engine = Engine(cfg, self.runtimes[0], self.tokenizer, self.kv_blocks, board=self.board,
                             tracer=self.tracer).start()

    @staticmethod
    def _run_stage(rt):
        try:
            stage_main(rt)
        except BaseException:
            pass

    def close(self):
        self.engine.stop()
        if self.engine.fatal is not None and len(self.runtimes) > 1:
            # after a fatal error the engine does not drain its stages (they may be the ones that hang); stage
            # THREADS that are still responsive are told to stop all the same, or they would outlive the pipeline
            try:
                from ..runtime.batch import KIND_SHUTDOWN, control
                self.runtimes[0].links.next_ctrl.put(control(KIND_SHUTDOWN).to_bytes())
            except Exception:
                pass
        for t in self.threads:
            t.join(5.0)
        self.runtimes[0].dump_trace()
        if not any(t.is_alive() for t in self.threads):
            for rt in self.runtimes:                  # device hops and rings of an in-process GPU pipeline
                rt.links.close()
            if self._own_executors and self.engine.fatal is None:
                for ex in self.executors:             # graphs and static buffers go now, on this thread
                    if hasattr(ex, "drop_graphs"):
                        ex.drop_graphs()
        self.board.close()


# ---------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_distributed(rank: int, world: int, backend: str, master_port: int, device: Optional[str]):
    """Default process group of a multi-process pipeline. With the native RCCL transport it is gloo:
    it carries communicator ids, barriers and the benchmark's max-over-ranks, never activations."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(master_port)       # not setdefault: a rebuilt pipeline gets a new port
    if device is not None and device.startswith("cuda"):
        torch.cuda.set_device(torch.device(device))
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=600))
    return dist


def rccl_links_or_fallback(rank: int, world: int, job: str, device: str, slot: int,
                           warmup_timeout_s: float = 120.0):
    """The stage's links over the native RCCL hops — or, when any rank fails to bring them up, over
    the next transport that every rank can: `torch.distributed` NCCL point-to-point groups, then gloo
    with activations staged through pinned host memory. The ranks agree after every attempt (a MIN
    over the gloo default group), so they always end up on the same transport; a failed attempt's
    hops are aborted and closed first. `links.transport` names what is in use. A pipeline that comes
    up slower beats one that does not come up (the reference's answer to an unreachable worker is an
    error string per request: orchestration.py:118-122)."""
    import torch.distributed as dist
    from ..utils.tracing import get_logger
    from .transport import make_dist_links, make_rccl_links
    attempts = [
        ("rccl", lambda j: make_rccl_links(rank, world, j, device, ctrl_slot_bytes=slot, warmup_timeout_s=warmup_timeout_s)),
        ("rccl-torch", lambda j: make_dist_links(rank, world, "nccl", j, host_staged=False, ctrl_slot_bytes=slot)),
        ("gloo", lambda j: make_dist_links(rank, world, "gloo", j, host_staged=True, ctrl_slot_bytes=slot)),
    ]
    log = get_logger()
    last = None
    bringup_timeout_s = float(os.environ.get("DLIA_TRANSPORT_BRINGUP_TIMEOUT_S", "180"))
    for i, (name, make) in enumerate(attempts):
        links, err = None, None
        # Communicator creation blocks inside the library until every peer has joined; a fabric that
        # never lets them meet would hang the rank for good. The attempt therefore runs on a helper
        # thread with a deadline: a rank that gives up reports failure like any other, all ranks move
        # on to the next transport together, and the stuck thread is left behind (daemon).
        box = {}

        def attempt(make=make, j=job if i == 0 else f"{job}_t{i}"):      # fresh ring names per attempt
            try:
                if device.startswith("cuda"):
                    torch.cuda.set_device(torch.device(device))          # HIP's current device is per thread
                box["links"] = make(j)
            except BaseException as e:                            # noqa: BLE001 - reported below, agreed on by all ranks
                box["err"] = e

        th = threading.Thread(target=attempt, name=f"dlia-bringup-{name}", daemon=True)
        th.start()
        th.join(bringup_timeout_s if name != "gloo" else None)
        if th.is_alive():
            err = TimeoutError(f"{name} links did not come up within {bringup_timeout_s:.0f} s")
        else:
            links, err = box.get("links"), box.get("err")
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            links.transport = name
            if i > 0:
                log.warning("stage %d: pipeline links are on the %s transport (fallback %d)", rank, name, i)
            return links
        if err is not None:
            log.warning("stage %d: %s transport did not come up: %s: %s", rank, name, type(err).__name__, err)
            last = err
        if links is not None:
            links.abort()
            links.close()
    raise RuntimeError(f"no stage-to-stage transport could be brought up on every rank (last error: {last})")


def _group_backend(transport: str) -> str:
    """torch.distributed backend of the default group for a transport: the native RCCL hops and
    gloo both use a gloo group; "rccl-torch" moves activations through ProcessGroupNCCL itself."""
    return {"rccl": "gloo", "gloo": "gloo", "rccl-torch": "nccl", "nccl": "nccl"}.get(transport, "gloo")


def stage_process_setup(cfg: EngineConfig, rank: int, world: int, job: str, master_port: int,
                        local_rank: Optional[int] = None, create_board: Optional[bool] = None):
    """Common bring-up of one stage process. Returns (spec, plan, executor, links, board).
    `create_board`: whether this process creates the status board (default: rank 0 does; the
    stages of a `SpawnedPipeline` all attach to the one their parent created)."""
    dev_kind = cfg.resolved_device()
    transport = cfg.resolved_transport()
    backend = _group_backend(transport)
    device, share = "cpu", 1.0
    if dev_kind == "cuda":
        idx = rank if local_rank is None else local_rank
        ndev = torch.cuda.device_count()
        if world > ndev > 0 and (transport == "gloo" or os.environ.get("DLIA_SHARE_GPU") == "1"):
            # more stages than GPUs: the host-staged transport can share a device between ranks, and so can
            # the RCCL transports in rehearsal mode (below); used to rehearse a multi-stage job on a small box
            share = 1.0 / ((world + ndev - 1) // ndev)
            idx %= ndev
            if transport != "gloo" and os.environ.get("DLIA_RCCL_DISTINCT_HOSTS") == "1":
                # RCCL refuses two ranks of one communicator on one device of one HOST, and tells hosts
                # apart by NCCL_HOSTID. With a different id per rank the ranks count as different hosts and
                # their hops go over RCCL's socket transport — the same ncclSend / ncclRecv calls, fences and
                # streams as on a full node, minus xGMI. Read by RCCL when it first initialises.
                os.environ["NCCL_HOSTID"] = f"dlia-rehearsal-rank-{rank}"
        device = f"cuda:{idx}"
        torch.cuda.set_device(torch.device(device))
    spec = resolve_spec(cfg)
    plan = plans_for(cfg, spec)[rank]
    board = StatusBoard(job + "_b", create=(rank == 0) if create_board is None else create_board, num_stages=world)
    board.register(rank, plan.layer_start, plan.layer_end)
    try:
        if world > 1:
            if dev_kind == "cuda":
                from .rccl import limit_p2p_channels
                limit_p2p_channels()             # before any RCCL communicator exists (also the rccl-torch fallback's)
            init_distributed(rank, world, backend, master_port, device if dev_kind == "cuda" else None)
        ex = build_executor(cfg, spec, plan, device=device, defer_kv=True)
        settle_kv(ex, board.agree_kv_blocks(rank, kv_budget(cfg, ex, share)), cfg.warm_decode_graphs)
        slot = ctrl_slot_bytes(cfg)
        if world == 1:
            links = make_loopback_links(1)[0]
        elif transport == "rccl" and dev_kind == "cuda":
            links = rccl_links_or_fallback(rank, world, job, device, slot)
        else:
            links = make_dist_links(rank, world, backend, job, host_staged=(dev_kind == "cuda" and backend == "gloo"),
                                    ctrl_slot_bytes=slot)
    except BaseException as e:
        board.set_state(rank, ST_ERROR, f"{type(e).__name__}: {e}")
        raise
    return spec, plan, ex, links, board


def open_links_only(cfg: EngineConfig, rank: int, world: int):
    """The default group and this rank's pipeline hops, without a model or a KV cache: what a
    pre-flight check of a node opens, times and closes (`bench.py --preflight`). Reads MASTER_PORT /
    LOCAL_RANK like `attach_distributed`. Returns (links, device or None on CPU)."""
    dev_kind = cfg.resolved_device()
    transport = cfg.resolved_transport()
    backend = _group_backend(transport)
    port = int(os.environ.get("MASTER_PORT", "29500"))
    job = f"dlia_pf_{port}_{os.getppid()}"
    device = None
    if dev_kind == "cuda":
        idx = int(os.environ.get("LOCAL_RANK", str(rank)))
        ndev = torch.cuda.device_count()
        if world > ndev > 0 and (transport == "gloo" or os.environ.get("DLIA_SHARE_GPU") == "1"):
            idx %= ndev
            if transport != "gloo" and os.environ.get("DLIA_RCCL_DISTINCT_HOSTS") == "1":
                os.environ["NCCL_HOSTID"] = f"dlia-rehearsal-rank-{rank}"
        device = f"cuda:{idx}"
        torch.cuda.set_device(torch.device(device))
        from .rccl import limit_p2p_channels
        limit_p2p_channels()
    init_distributed(rank, world, backend, port, device)
    if transport == "rccl" and dev_kind == "cuda":
        links = rccl_links_or_fallback(rank, world, job, device, 1 << 16)
    else:
        links = make_dist_links(rank, world, backend, job, host_staged=(dev_kind == "cuda" and backend == "gloo"),
                                ctrl_slot_bytes=1 << 16)
    return links, device


def _exit_with_parent(job: str, poll_s: float = 1.0):
    """A stage process must not outlive the serving process that started it: were that one killed
    outright (out of memory, `kill -9`) its stages would keep their GPUs and spin on rings nobody
    feeds. A daemon thread watches the parent pid; when it changes it removes the job's shared-memory
    objects (nobody else will) and ends this process."""
    import threading
    parent = os.getppid()

    def watch():
        while os.getppid() == parent:
            time.sleep(poll_s)
        import glob
        for path in glob.glob(f"/dev/shm/{job}_*"):       # the parent can no longer sweep; mappings stay valid for siblings
            try:
                os.unlink(path)
            except OSError:
                pass
        os._exit(1)

    threading.Thread(target=watch, name="dlia-parent-watch", daemon=True).start()


def _stage_entry(cfg_dict: dict, rank: int, world: int, job: str, master_port: int):
    """Body of one spawned stage process: stage 0 serves the scheduler's descriptor ring
    (`head_main`), every other stage the ring of its predecessor (`stage_main`)."""
    from ..runtime.head import head_main
    from .transport import ShmControl
    _exit_with_parent(job)
    cfg = EngineConfig.from_dict(cfg_dict)
    spec, plan, ex, links, board = stage_process_setup(cfg, rank, world, job, master_port, create_board=False)
    rt = StageRuntime(ex, links, board=board, fault=cfg.fault_inject, tracer=make_tracer(cfg.trace_path),
                      decode_streams=cfg.decode_streams, debug_top5=cfg.debug_top5, trace_path=cfg.trace_path)
    try:
        if rank == 0:
            cin = ShmControl(f"{job}_cin", create=False)
            res = ShmControl(f"{job}_res", create=False)
            head_main(rt, cin, res)
        else:
            stage_main(rt)
    finally:
        links.close()


class SpawnedPipeline:
    """One OS process per stage (rank = stage = GPU), all of them children of this process, which
    keeps only the scheduler: it holds no device context and no communicator, reaches stage 0
    through two shared-memory rings, and can therefore always kill and replace the stages — also
    when one of them sits in a device call that will never return."""

    def __init__(self, cfg: EngineConfig, ready_timeout: float = 600.0):
        import multiprocessing as mp
        from ..runtime.engine import Engine
        from ..runtime.head import RemoteHead
        from .transport import ShmControl
        self.cfg = cfg
        world = cfg.pp_size
        ctx = mp.get_context("spawn")
        self.spec = resolve_spec(cfg)
        self.plans = plans_for(cfg, self.spec)
        self.engine = None
        for attempt in range(3):
            self.job = f"dlia_{os.getpid()}_{uuid.uuid4().hex[:8]}"
            port = _free_port()
            # board and scheduler rings exist before the children attach
            self.board = StatusBoard(self.job + "_b", create=True, num_stages=world)
            slot = ctrl_slot_bytes(cfg)
            self.cin = ShmControl(f"{self.job}_cin", create=True, slot_bytes=slot)
            self.res = ShmControl(f"{self.job}_res", create=True, slot_bytes=max(1 << 14, 64 + 4 * cfg.max_batch + 2048))
            self.procs = []
            # (the children learn from the environment they inherit that they are stages of a pipeline of `world` ranks:
            # a stage of several that owns its GPU wants more hardware queues than a lone one, see the package's __init__)
            prev_ranks = os.environ.get("DLIA_PP_RANKS")
            os.environ["DLIA_PP_RANKS"] = str(world)
            try:
                for r in range(world):
                    p = ctx.Process(target=_stage_entry, args=(cfg.to_dict(), r, world, self.job, port), daemon=True)
                    p.start()
                    self.procs.append(p)
            finally:
                if prev_ranks is None:
                    os.environ.pop("DLIA_PP_RANKS", None)
                else:
                    os.environ["DLIA_PP_RANKS"] = prev_ranks
            try:
                self.wait_ready(ready_timeout)
                break
            except BaseException as e:
                # A failed bring-up leaves nothing behind: the caller gets no object to close, so the stage
                # processes are killed and every shared-memory object of this attempt (scheduler rings,
                # status board, the rings the stages created for their successors) is removed here.
                self.kill()
                for obj in (self.cin, self.res, self.board):
                    try:
                        obj.close()
                    except Exception:
                        pass
                self._sweep_shm()
                # The rendezvous port was free when picked, but another socket of this machine may take it
                # before rank 0 listens on it: that attempt is discarded whole and made again on a new port.
                clash = isinstance(e, RuntimeError) and ("failed to listen" in str(e) or "ddress already in use" in str(e))
                if not clash or attempt == 2:
                    raise
        self.kv_blocks = min(self.board.read(s)["kv_blocks"] for s in range(world))
        hip = cfg.resolved_backend() == "hip"
        head = RemoteHead(self.spec, world, self.cin, self.res, graph_capable=hip, graph_rows=cfg.graph_sizes(),
                          debug_top5=cfg.debug_top5, timeout_s=cfg.hop_timeout_s * world)
        self.tracer = make_tracer(cfg.trace_path)
        self.tokenizer = make_tokenizer(cfg, self.spec)
        self.engine = Engine(cfg, head, self.tokenizer, self.kv_blocks, board=self.board, tracer=self.tracer).start()

    def wait_ready(self, timeout: float = 300.0):
        t0 = time.time()
        while time.time() - t0 < timeout:
            recs = [self.board.read(s) for s in range(self.cfg.pp_size)]
            if all(r["state"] == "ready" for r in recs):
                return True
            dead = [i for i, p in enumerate(self.procs) if not p.is_alive()]
            if any(r["state"] == "error" for r in recs) or dead:
                raise RuntimeError(f"stage bring-up failed (exited: {dead}): {recs}")
            time.sleep(0.05)
        raise TimeoutError("pipeline stages did not come up")

    def kill(self):
        """Replace-don't-repair: SIGKILL every stage process. The driver tears their device contexts
        and communicators down; nothing in this process waits on a device."""
        for p in self.procs:
            if p.is_alive():
                p.kill()
        for p in self.procs:
            p.join(10.0)

    def close(self):
        fatal = self.engine is not None and self.engine.fatal is not None
        if self.engine is not None:
            self.engine.stop()
        if fatal:
            self.kill()
        else:
            for p in self.procs:
                p.join(15.0)
            self.kill()
        if self.tracer is not None and self.cfg.trace_path:
            # stage 0 wrote `trace_path` (with the other stages' files folded in): add the scheduler's events
            try:
                path = self.cfg.trace_path
                if os.path.exists(path):
                    os.replace(path, path + ".stage0")
                self.tracer.dump(path, merge_glob=path + ".stage*")
            except Exception:
                pass
        for ring in (self.cin, self.res):
            try:
                ring.close()
            except Exception:
                pass
        self.board.close()
        self._sweep_shm()

    def _sweep_shm(self):
        """Every shared-memory object of this job carries its unique name prefix. A stage process that was
        killed, or died, never unlinked the descriptor ring it had created for its successor (tens of
        megabytes of memory each, held until reboot): once all stage processes are gone the parent removes
        whatever is left under the prefix."""
        import glob
        if any(p.is_alive() for p in self.procs):
            return
        for path in glob.glob(f"/dev/shm/{self.job}_*"):
            try:
                os.unlink(path)
            except OSError:
                pass


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
    """The right pipeline object for `cfg`: in-process for the loopback transport (and for one stage
    unless `isolate` asks for a separate stage process), spawned stage processes otherwise."""
    if cfg.resolved_transport() == "loopback" and not cfg.isolate:
        return LocalPipeline(cfg)
    return SpawnedPipeline(cfg)
