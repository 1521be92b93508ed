"""HTTP stage worker with the reference's wire contract (`/`, `/health`, `/process`).

This is the plumbing-parity transport: hidden states as JSON nested lists over HTTP, exactly
what the reference's workers speak (Worker1.py:185-245). A request with only `hidden_states`
is processed statelessly — positions start at 0 and nothing is cached, the reference's
behaviour (Worker1.py:93-94). Two optional fields make the same endpoint incremental:
`seq_id` (which sequence's KV cache to use) and `ctx_len` (how many tokens are already cached),
so an orchestrator that knows about them sends only the new tokens' rows. `POST /release`
drops a sequence's cache. The fast path of this engine is the RCCL pipeline
(`parallel/pipeline.py`), not this.
"""
from __future__ import annotations

import threading
import traceback
from typing import Optional

import torch

from ..config import EngineConfig
from ..models.partition import StagePlan
from ..runtime.batch import StepBatch


class WorkerService:
    """Thread-safe wrapper around one layers-only stage executor."""

    def __init__(self, executor, worker_name: str, model_name: str):
        self.ex = executor
        self.worker_name = worker_name          # e.g. "Worker 1" (Worker1.py:29)
        self.model_name = model_name
        self.plan: StagePlan = executor.plan
        self._lock = threading.Lock()
        self._scratch_slot = 1 << 20            # stateless requests run in a slot of their own
        self._slots = {}                        # seq_id -> slot
        self._next_slot = 0
        self.requests = 0

    @property
    def role(self) -> str:
        return self.worker_name.lower().replace(" ", "_")      # Worker1.py:203

    def process(self, hidden_states: torch.Tensor, seq_id: Optional[str] = None, ctx_len: int = 0) -> torch.Tensor:
        """hidden_states [B, S, H] -> [B, S, H]."""
        if hidden_states.dim() != 3 or hidden_states.shape[-1] != self.ex.spec.hidden_size:
            raise ValueError(f"hidden_states must be [batch, seq, {self.ex.spec.hidden_size}], got {tuple(hidden_states.shape)}")
        B, S, H = hidden_states.shape
        with self._lock:
            self.requests += 1
            if seq_id is None:
                slots = [self._scratch_slot + b for b in range(B)]
                ctx = [0] * B
            else:
                slots = []
                for b in range(B):
                    key = f"{seq_id}/{b}"
                    if key not in self._slots:
                        self._slots[key] = self._next_slot
                        self._next_slot += 1
                    slots.append(self._slots[key])
                ctx = [int(ctx_len)] * B
            n = len(slots)
            batch = StepBatch(slots=slots, q_lens=[S] * n, ctx_lens=ctx, sample=[0] * n, temperature=[0.0] * n,
                              top_k=[0] * n, top_p=[1.0] * n, seeds=[0] * n, sample_steps=[0] * n)
            try:
                out = self.ex.forward(batch, hidden_states.reshape(B * S, H))
            finally:
                if seq_id is None:
                    self.ex.release(slots)
            return out.reshape(B, S, H).float().cpu()

    def release(self, seq_id: str) -> int:
        with self._lock:
            keys = [k for k in self._slots if k.split("/")[0] == str(seq_id)]
            self.ex.release([self._slots.pop(k) for k in keys])
            return len(keys)


def create_worker_app(service: Optional[WorkerService]):
    from flask import Flask, jsonify, request
    app = Flask(f"dlia-{service.role if service else 'worker'}")
    app.config["service"] = service

    @app.after_request
    def _cors(resp):                                  # the reference enables CORS (Worker1.py:181)
        resp.headers["Access-Control-Allow-Origin"] = "*"
        resp.headers["Access-Control-Allow-Headers"] = "Content-Type, ngrok-skip-browser-warning"
        return resp

    @app.route("/")
    def home():
        s = app.config["service"]
        if s is None:
            return "<html><body><h1>worker not initialized</h1></body></html>"
        p = s.plan
        # a status page like the reference's worker has (Worker1.py:185-197), own markup: a table of facts
        facts = [("stage", s.worker_name), ("decoder layers", f"{p.layer_start} to {p.layer_end} ({p.num_layers})"),
                 ("model", s.model_name), ("executor", s.ex.backend), ("hidden-state requests served", s.requests)]
        rows = "".join(f"<tr><th>{k}</th><td>{v}</td></tr>" for k, v in facts)
        return ("<!DOCTYPE html><html><head><meta charset=\"utf-8\"><title>pipeline stage</title>"
                "<style>body{font:15px/1.5 system-ui,sans-serif;margin:3rem auto;max-width:36rem;color:#1c2733}"
                "table{border-collapse:collapse;width:100%}th,td{text-align:left;padding:.35rem .6rem;border-bottom:1px solid #d5dbe1}"
                "th{width:45%;font-weight:600}.up{color:#0a7d38}</style></head>"
                f"<body><h1>Pipeline stage <span class=\"up\">up</span></h1><table>{rows}</table></body></html>")

    @app.route("/health", methods=["GET"])
    def health():
        s = app.config["service"]
        if s is None:
            return jsonify({"error": "Worker not initialized"}), 500
        return jsonify({"status": "healthy", "role": s.role, "layers": s.plan.describe(), "model": s.model_name})

    @app.route("/process", methods=["POST"])
    def process():
        s = app.config["service"]
        if s is None:
            return jsonify({"error": "Worker not initialized"}), 500          # Worker1.py:212-213
        try:
            data = request.get_json(silent=True) or {}
            hs = data.get("hidden_states")
            if not hs:
                return jsonify({"error": "No hidden states provided"}), 400   # Worker1.py:219-220
            x = torch.tensor(hs, dtype=torch.float32)
            out = s.process(x, seq_id=data.get("seq_id"), ctx_len=int(data.get("ctx_len", 0)))
            return jsonify({"hidden_states": out.tolist(), "status": "success", "worker": s.role})
        except Exception as e:
            traceback.print_exc()
            return jsonify({"error": f"Error: {e}"}), 500                     # Worker1.py:241-245

    @app.route("/release", methods=["POST"])
    def release():
        s = app.config["service"]
        if s is None:
            return jsonify({"error": "Worker not initialized"}), 500
        data = request.get_json(silent=True) or {}
        return jsonify({"released": s.release(str(data.get("seq_id", ""))), "status": "success"})

    return app


def build_worker_service(cfg: EngineConfig, worker_index: int, num_workers: int) -> WorkerService:
    """Worker `worker_index` (0-based) of `num_workers`: loads ONLY its own layers."""
    from ..models.partition import make_worker_plans
    from ..models.stage_ref import RefStage
    from ..parallel.launcher import load_weights, resolve_spec
    spec = resolve_spec(cfg)
    _, plans = make_worker_plans(spec, num_workers, mode=cfg.partition, ranges=cfg.layer_ranges)
    plan = plans[worker_index]
    if cfg.resolved_backend() == "hip":
        from ..models.stage_gpu import GpuStage
        w = load_weights(cfg, spec, plan, cfg.resolved_device(), torch.bfloat16)
        ex = GpuStage(spec, plan, w, device=cfg.resolved_device(), max_slots=cfg.max_batch * 4,
                      max_seq_len=cfg.max_seq_len, max_tokens=cfg.prefill_chunk, max_batch=cfg.max_batch,
                      use_graphs=False)
        ex = _GpuWorkerAdapter(ex)
    else:
        ex = RefStage(spec, plan, load_weights(cfg, spec, plan, "cpu", torch.float32), dtype=torch.float32,
                      max_seq_len=cfg.max_seq_len)
    return WorkerService(ex, f"Worker {worker_index + 1}", spec.name)


class _GpuWorkerAdapter:
    """Gives a GpuStage the slot/block bookkeeping the HTTP worker needs (it has no scheduler
    above it): every slot gets a fixed run of blocks on first use."""

    def __init__(self, stage):
        self.stage = stage
        self.spec, self.plan, self.backend = stage.spec, stage.plan, stage.backend
        self._map = {}
        self._free = list(range(stage.max_slots))

    def forward(self, batch: StepBatch, x: torch.Tensor) -> torch.Tensor:
        per = self.stage.max_blocks
        for i, s in enumerate(batch.slots):
            if s not in self._map:
                if not self._free:
                    raise RuntimeError("worker out of KV slots")
                local = self._free.pop()
                self._map[s] = local
                first = local * per
                if first + per > self.stage.num_blocks:
                    raise RuntimeError("worker out of KV blocks")
                batch.table_slots.append(local)
                batch.table_blocks.append(list(range(first, first + per)))
            batch.slots[i] = self._map[s]
        return self.stage.forward(batch, x).clone()

    def release(self, slots):
        for s in slots:
            local = self._map.pop(s, None)
            if local is not None:
                self._free.append(local)


def serve_worker(cfg: EngineConfig, worker_index: int, num_workers: int, port: Optional[int] = None):
    """Blocking: build the worker and serve it (default port 5001 + index: Worker1.py:30, Worker2.py:29)."""
    service = build_worker_service(cfg, worker_index, num_workers)
    app = create_worker_app(service)
    app.run(host=cfg.host, port=port or (5001 + worker_index), debug=False, threaded=True)
