"""Typed engine configuration (dataclass <- YAML / CLI / environment).

The reference has no config system: model name, worker URLs, layer ranges, ports and the ngrok
token are module constants edited by hand before running (orchestration.py:20-24;
Worker1.py:26-31), `top_k`/`top_p` are literals in the route (orchestration.py:353-354), the
generation cap is a literal 30 (orchestration.py:347) and the number of workers (2) is baked
into the control flow. Everything is a field here, and the pipeline is N-stage generic.
"""
from __future__ import annotations

import dataclasses
import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple


@dataclass
class EngineConfig:
    # ---- model -------------------------------------------------------------------------
    model: str = "tinyllama-1.1b"          # built-in name or HF checkpoint directory
    weights: str = "auto"                  # "auto": checkpoint if `model` is a dir with weights, else random
    dtype: str = "bf16"                    # bf16 | fp8 (e4m3 weights x e4m3 activations on the fp8 MFMA) | fp8w (e4m3 weights only)
    kv_dtype: str = "bf16"                 # bf16 | fp8 (e4m3 KV cache with one scale per token and kv head; HIP backend)
    seed: int = 0                          # random-init seed
    tokenizer: Optional[str] = None        # tokenizer directory (default: the model directory; else byte fallback)
    # ---- pipeline ----------------------------------------------------------------------
    pp_size: int = 1                       # number of pipeline stages (= processes = GPUs)
    partition: str = "balanced"            # "balanced" | "even" | "explicit"
    split_layers: bool = True              # balanced: a stage boundary may fall between a layer's attention and MLP blocks
    layer_ranges: Optional[List[Tuple[int, int]]] = None
    backend: str = "auto"                  # stage executor: "hip" | "torch" | "auto"
    device: str = "auto"                   # "cuda" | "cpu" | "auto"
    transport: str = "auto"                # "rccl" (native hops) | "rccl-torch" | "gloo" | "loopback" | "http" | "auto"
    isolate: bool = False                  # run even a single stage in its own process (the scheduler's process then holds no GPU context)
    # ---- batching / memory ------------------------------------------------------------
    max_batch: int = 64                    # sequences per micro-batch
    num_micro_batches: int = 0             # in-flight micro-batches (0 -> pp_size, >= 1)
    max_seq_len: int = 2048
    # max new tokens per prefill step. 4096 rows fill the 256-CU chip with whole rounds of 256 x 256 GEMM
    # tiles (Llama-3-8B gate/up: 1792 tiles = 7.0 rounds; 2048 rows: 3.5): prefill alone 91 k tok/s against
    # 82 k, a 4096-request serving run 363 against 339 requests/s with lower TTFT and TPOT (one MI355X)
    prefill_chunk: int = 4096
    # A prefill step of a micro-batch also carries one decode token for each of its running sequences
    # (they ride in the same ragged step instead of standing still while newcomers are prefilled)
    mixed_prefill: bool = True
    kv_blocks: int = 0                     # KV blocks per stage (0 -> sized from free device memory, see kv_mem_fraction)
    kv_mem_fraction: float = 0.85          # share of the memory left after weights and workspaces that the KV cache takes
    use_graphs: bool = True                # hipGraph-captured decode
    # Row counts (besides max_batch) for which a micro-batch keeps a captured decode step: with few
    # sequences running, a step covers only the rows up to the highest one in use, rounded up to the
    # next of these, instead of all max_batch rows (a single request on a 256-row micro-batch:
    # the step of 8 rows instead of 256). [] = every decode step covers max_batch rows; `serve` on the
    # HIP backend fills in 8, 64, 512 (those below max_batch) when max_batch >= 128 and none are given.
    decode_buckets: List[int] = field(default_factory=list)
    warm_decode_graphs: bool = True        # with decode_buckets: allocate and capture every (micro-batch, size) step at bring-up rather than at first use
    max_inflight_steps: int = 2            # steps of one micro-batch issued ahead of their results
    # A running micro-batch stops to take waiting requests in (drain, prefill, restart its decode loop)
    # only when that fills at least this share of its rows, or the oldest waiter has waited this long:
    # (0 / 0: stop for any waiter at once. Closed-loop load of 512 clients on one MI355X measured the
    # same 360 requests/s at 0 / 0, 0.125 / 25 ms and 0.25 / 50 ms, so the latency-friendly setting is
    # the default; raise them when arrivals are sparse and stops dominate.)
    admit_fill_fraction: float = 0.0
    admit_max_wait_ms: float = 0.0
    decode_streams: int = 3                # HIP streams the decode micro-batches of a stage alternate over
    # ---- sampling defaults (reference values) -----------------------------------------
    temperature: float = 0.7               # orchestration.py:341
    top_k: int = 50                        # orchestration.py:353
    top_p: float = 0.9                     # orchestration.py:354
    default_max_tokens: int = 20           # orchestration.py:340
    max_tokens_cap: int = 512              # the reference clamps to 30 (orchestration.py:347)
    system_prompt: str = "You are a helpful assistant."   # orchestration.py:66
    chat_template: str = "auto"            # "zephyr" (reference) | "hf" | "none" | "auto"
    # ---- serving ------------------------------------------------------------------------
    host: str = "0.0.0.0"                  # orchestration.py:372
    port: int = 5000                       # orchestration.py:372; workers use 5001.. (Worker1.py:30)
    server: str = "uvicorn"                # HTTP server of `serve`: "uvicorn" (ASGI) | "flask" (Werkzeug dev server, as the reference)
    worker_urls: List[str] = field(default_factory=list)   # http transport: one URL per worker
    request_timeout_s: float = 200.0       # Test.py:71
    shutdown_grace_s: float = 30.0         # SIGTERM: requests still open after this long are cancelled and the server leaves
    max_queue: int = 0                     # requests allowed to wait for a row; 0 = unbounded. Beyond it a request fails at once ("queue full")
    stream_cost_ms: float = 0.25           # streaming: a stream sends at most one event per (open streams x this)
    hop_timeout_s: float = 30.0            # orchestration.py:118
    health_timeout_s: float = 5.0          # orchestration.py:313
    debug_top5: bool = False               # orchestration.py:172-178
    trace_path: Optional[str] = None       # write a Chrome trace of step timelines here
    auto_restart: bool = False             # rebuild the pipeline (new processes, new communicators) after a fatal stage failure
    # ---- fault injection (tests) ------------------------------------------------------
    fault_inject: str = ""                 # e.g. "stage1:exit@3", "stage1:stall@2:5.0", "stage0:raise@1"

    def graph_sizes(self) -> List[int]:
        """Ascending row counts of fixed-shape decode steps: the buckets below max_batch, then max_batch."""
        return sorted({int(b) for b in self.decode_buckets if 0 < int(b) < self.max_batch}) + [self.max_batch]

    def __post_init__(self):
        if self.pp_size < 1:
            raise ValueError("pp_size must be >= 1")
        if self.num_micro_batches <= 0:
            self.num_micro_batches = max(1, self.pp_size)
        if self.layer_ranges is not None:
            self.layer_ranges = [tuple(r) for r in self.layer_ranges]
            self.partition = "explicit"
        if self.dtype not in ("bf16", "fp32", "fp8", "fp8w"):
            raise ValueError(f"unsupported dtype {self.dtype!r}")
        if self.kv_dtype not in ("bf16", "fp8"):
            raise ValueError(f"unsupported kv_dtype {self.kv_dtype!r}")
        if self.max_batch < 1 or self.max_seq_len < 8:
            raise ValueError("max_batch / max_seq_len out of range")

    # ---- construction helpers ---------------------------------------------------------
    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "EngineConfig":
        names = {f.name for f in dataclasses.fields(cls)}
        unknown = set(d) - names
        if unknown:
            raise ValueError(f"unknown config keys: {sorted(unknown)}")
        return cls(**d)

    @classmethod
    def from_yaml(cls, path: str, **overrides) -> "EngineConfig":
        import yaml
        with open(path) as f:
            d = yaml.safe_load(f) or {}
        d.update({k: v for k, v in overrides.items() if v is not None})
        return cls.from_dict(d)

    @classmethod
    def from_env(cls, base: Optional["EngineConfig"] = None, prefix: str = "DLIA_") -> "EngineConfig":
        """Environment overrides: DLIA_PP_SIZE=4, DLIA_MODEL=llama-3-8b, ..."""
        d = dataclasses.asdict(base) if base else {}
        for f in dataclasses.fields(cls):
            v = os.environ.get(prefix + f.name.upper())
            if v is None:
                continue
            d[f.name] = _coerce(v, f.type)
        return cls.from_dict(d)

    def to_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)

    def resolved_device(self) -> str:
        if self.device != "auto":
            return self.device
        import torch
        # counting devices does not create a device context; the orchestrator process must not hold one
        return "cuda" if torch.cuda.device_count() > 0 else "cpu"

    def resolved_backend(self) -> str:
        if self.backend != "auto":
            return self.backend
        return "hip" if self.resolved_device() == "cuda" else "torch"

    def resolved_transport(self) -> str:
        if self.transport != "auto":
            return self.transport
        if self.pp_size == 1:
            return "loopback"
        return "rccl" if self.resolved_device() == "cuda" else "gloo"


def _coerce(v: str, typ) -> Any:
    t = str(typ)
    if "bool" in t:
        return v.lower() in ("1", "true", "yes", "on")
    if "int" in t and "List" not in t and "Tuple" not in t:
        return int(v)
    if "float" in t:
        return float(v)
    if "List[str]" in t:
        return [s for s in v.split(",") if s]
    if "List[int]" in t:
        return [int(s) for s in v.replace("[", "").replace("]", "").split(",") if s.strip()]     # "8,64" or "[8, 64]"
    if "Tuple" in t:
        # "0-11,11-22"
        return [tuple(int(x) for x in part.split("-")) for part in v.split(",") if part]
    return v
