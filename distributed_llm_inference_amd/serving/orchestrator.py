"""Orchestrator HTTP API with the reference's contract (orchestration.py:230-356):

    GET  /          HTML status page listing the workers / stages
    GET  /health    {"status":"healthy","role":"orchestrator","model":...,"version":...}
    GET  /workers   {"worker_1":"online"|"error"|"offline"|"not_configured", ...}
    POST /generate  {"prompt": str, "max_tokens": int = 20, "temperature": float = 0.7}
                    (additions, all optional: "top_k": int, "top_p": float, "seed": int, "stream": bool)
                    -> 200 {"prompt","response","status":"success","time_taken":"N.NNs",
                            "tokens_generated":int,"tokens_per_sec":"N.NN"}
                    with "stream": true (addition, ASGI server): text/event-stream of {"text","tokens_generated"}
                    events, then that dictionary, then [DONE]
                       400 {"error":"No prompt provided"}; 500 {"error":"Orchestrator not initialized"};
                       200 {"error":"Error: ...","status":"failed"} on a runtime failure
    GET  /metrics   Prometheus text (addition)
    GET  /ready     200 {"ready": true, ...} when every stage is online and the engine has not failed, else 503 (addition)
    /v1/models, /v1/completions, /v1/chat/completions   OpenAI-style routes (addition, ASGI server: openai_api.py)

Two backends sit behind the same routes: `EngineBackend` (the pipeline engine: stages over
RCCL / gloo / loopback, continuous batching) and `HttpRelayBackend` (the reference's topology —
embed / norm / lm_head / sampling here, layer workers behind `/process` URLs — kept for plumbing
parity). Handlers only enqueue and wait; all serving state lives in the engine's scheduler thread.
"""
from __future__ import annotations

import contextlib
import itertools
import os
import threading
import time
from typing import Dict, List

import torch

from ..config import EngineConfig
from ..runtime.sampler import sample_token, top_candidates
from ..runtime.tokenizer import IncrementalDetokenizer, build_prompt, encode_prompt

VERSION = "dlia-mi355x-0.1"
_HEADERS = {"ngrok-skip-browser-warning": "true", "Content-Type": "application/json"}   # orchestration.py:91-94


class EngineBackend:
    """/generate and /workers on top of a pipeline (`LocalPipeline` / `SpawnedPipeline`)."""

    def __init__(self, pipeline, factory=None):
        self.pipe = pipeline
        self.cfg: EngineConfig = pipeline.cfg
        self.engine = pipeline.engine
        self.model_name = pipeline.spec.name
        self.restarts = 0
        self._factory = factory
        self._restart_lock = threading.Lock()
        self._pipe_closed = False           # the current pipeline has been torn down for a rebuild that has not succeeded yet
        self.restart_failures = 0
        self._next_restart_at = 0.0
        self._streams = 0                   # open /generate streams (event-loop thread only)

    def _recover(self):
        """Elastic recovery: the reference needs a human to restart a dead worker and re-paste its
        URL (orchestration.py:22-24). With `auto_restart`, a pipeline whose engine went fatal (a
        stage died, reported an error or stalled) is torn down — processes, process groups,
        shared-memory rings — and rebuilt before the next request; requests that were in flight
        at the failure have already been answered with the error envelope."""
        if not (self.cfg.auto_restart and self.engine.fatal):
            return
        with self._restart_lock:
            if not self.engine.fatal:
                return
            # A bring-up that keeps failing (a GPU gone for good) must not be retried by every request:
            # attempts are spaced 1, 2, 4 ... 60 s apart; in between, and when an attempt fails, the
            # old engine stays in place with `fatal` set and answers with the error envelope.
            now = time.monotonic()
            if now < self._next_restart_at:
                return
            import dataclasses
            from ..parallel.launcher import make_pipeline
            from ..utils.tracing import get_logger
            if not self._pipe_closed:
                try:
                    self.pipe.close()
                except Exception:
                    pass
                self._pipe_closed = True
            cfg = dataclasses.replace(self.cfg, fault_inject="")      # injected faults are one-shot
            try:
                pipe = (self._factory or make_pipeline)(cfg)
            except BaseException as e:              # noqa: BLE001 - the request that triggered the attempt is answered, not crashed
                self.restart_failures += 1
                self._next_restart_at = time.monotonic() + min(60.0, 2.0 ** min(self.restart_failures - 1, 6))
                self.engine.fatal = f"{self.engine.fatal.split(' | restart failed')[0]} | restart failed: {type(e).__name__}: {e}"
                get_logger().error("pipeline restart %d failed: %s: %s", self.restart_failures, type(e).__name__, e)
                if not isinstance(e, Exception):
                    raise
                return
            self.pipe, self._pipe_closed = pipe, False
            self.cfg, self.engine = self.pipe.cfg, self.pipe.engine
            self.restarts += 1
            self.restart_failures, self._next_restart_at = 0, 0.0

    def generate(self, prompt, max_new_tokens, temperature, top_k=None, top_p=None, seed=None):
        self._recover()
        return self.engine.generate_text(prompt, max_new_tokens=max_new_tokens, temperature=temperature,
                                         top_k=self.cfg.top_k if top_k is None else top_k,
                                         top_p=self.cfg.top_p if top_p is None else top_p, seed=seed)

    def _sampling(self, top_k, top_p):
        return (self.cfg.top_k if top_k is None else top_k, self.cfg.top_p if top_p is None else top_p)

    async def recover_if_needed(self):
        import asyncio
        if self.cfg.auto_restart and self.engine.fatal:
            await asyncio.get_running_loop().run_in_executor(None, self._recover)   # rebuilding stage processes blocks

    async def wait_request(self, req, client_gone=None):
        """Waits on an event loop for a submitted request: the coroutine is resumed by the scheduler
        thread's completion callback, so thousands of requests can wait without a thread each.
        `client_gone`: optional coroutine function that returns once the client has disconnected.
        Returns None when the request is done; when its client left, or it outlived
        `request_timeout_s`, the request is cancelled in the engine (which frees its row and KV
        blocks instead of decoding to `max_tokens` for nobody) and the reason is returned. A handler
        task cancelled at server shutdown cancels the request too."""
        import asyncio
        loop = asyncio.get_running_loop()
        watch = None
        try:
            fut = loop.create_future()

            def wake(_req):
                loop.call_soon_threadsafe(lambda: fut.done() or fut.set_result(None))

            req.add_done_callback(wake)
            waiters = {fut}
            if client_gone is not None:
                watch = asyncio.ensure_future(client_gone())
                waiters.add(watch)
            done, _ = await asyncio.wait(waiters, timeout=self.cfg.request_timeout_s, return_when=asyncio.FIRST_COMPLETED)
            if fut in done:
                return None
            self.engine.cancel(req)
            return "client disconnected" if watch is not None and watch in done \
                else f"no result within {self.cfg.request_timeout_s:g}s"
        except asyncio.CancelledError:
            self.engine.cancel(req)
            raise
        finally:
            if watch is not None and not watch.done():
                watch.cancel()

    async def agenerate(self, prompt, max_new_tokens, temperature, client_gone=None, top_k=None, top_p=None, seed=None):
        """`generate` for an event loop (see `wait_request`)."""
        await self.recover_if_needed()
        start = time.time()
        engine = self.engine
        try:
            req = engine.submit_text(prompt, max_new_tokens, temperature, *self._sampling(top_k, top_p), seed)
            why = await self.wait_request(req, client_gone)
            if why is not None:
                return {"error": f"Error: {why}", "status": "failed"}
            return engine.text_result(req, prompt, start)
        except Exception as e:                                      # HTTP 200 envelope, as the reference
            return {"error": f"Error: {e or type(e).__name__}", "status": "failed"}

    async def pieces(self, req):
        """Async generator over a submitted request's text as it is decoded: yields (new text,
        tokens so far) once per wake-up with new tokens — every decode step while few streams are
        open, every few steps when there are thousands. The scheduler thread only pokes an
        `asyncio.Event`; the coroutine reads the request's output list itself, so no token is lost
        whatever the timing; text is decoded incrementally (tokens that end inside a multi-byte
        character are held back until it is complete). Ends when the request is done; raises
        TimeoutError (request cancelled) after `request_timeout_s`; closing the generator early (the
        client went away) cancels the request."""
        import asyncio
        loop = asyncio.get_running_loop()
        engine = self.engine
        news = asyncio.Event()
        poked = [False]

        def poke(_req):                     # scheduler thread: one cross-thread call per wake-up, not per token
            if not poked[0]:
                poked[0] = True
                loop.call_soon_threadsafe(news.set)

        req.token_callbacks.append(poke)
        req.add_done_callback(lambda _r: loop.call_soon_threadsafe(news.set))
        self._streams += 1
        try:
            deadline = loop.time() + self.cfg.request_timeout_s
            detok = IncrementalDetokenizer(engine.tok)
            sent_n, t_sent = 0, 0.0
            while True:
                news.clear()                    # before reading the state: a poke after the read wakes the wait below
                poked[0] = False
                done = req.done.is_set()
                n = len(req.output)
                if n > sent_n or done:
                    # the last pass always flushes: text held back behind an incomplete character must
                    # leave even when no further token arrives (EOS is not appended to the output)
                    text = detok.push(req.output[:n], final=done)
                    if text:
                        yield text, n
                    sent_n, t_sent = n, loop.time()
                if done:
                    return
                left = deadline - loop.time()
                if left <= 0:
                    raise TimeoutError(f"no result within {self.cfg.request_timeout_s:g}s")
                try:
                    await asyncio.wait_for(news.wait(), timeout=left)
                except asyncio.TimeoutError:
                    continue
                # One event loop serves every stream: the more of them, the longer each waits between
                # events (several tokens per event), so that the loop — and the scheduler thread it
                # shares the interpreter with — is not spent on per-token wake-ups.
                pause = t_sent + self._streams * self.cfg.stream_cost_ms * 1e-3 - loop.time()
                if pause > 0 and not req.done.is_set():
                    await asyncio.sleep(pause)
        finally:
            self._streams -= 1
            if not req.finished:
                engine.cancel(req)              # the deadline passed, or the client closed the stream

    async def astream(self, prompt, max_new_tokens, temperature, top_k=None, top_p=None, seed=None):
        """Server-sent events for `POST /generate` with `"stream": true`: one
        `data: {"text": <new text>, "tokens_generated": n}` event per piece (see `pieces`), then the
        reference's result dictionary (or its error envelope) as the last JSON event, then
        `data: [DONE]`. The reference answers only when the whole text is there
        (orchestration.py:201-218)."""
        import json
        await self.recover_if_needed()
        start = time.time()
        engine = self.engine

        def event(obj):
            return f"data: {json.dumps(obj)}\n\n"

        try:
            req = engine.submit_text(prompt, max_new_tokens, temperature, *self._sampling(top_k, top_p), seed)
            async with contextlib.aclosing(self.pieces(req)) as pieces:     # closed with this generator, not by the collector
                async for text, n in pieces:
                    yield event({"text": text, "tokens_generated": n})
            yield event(engine.text_result(req, prompt, start))
        except Exception as e:
            yield event({"error": f"Error: {e or type(e).__name__}", "status": "failed"})
        yield "data: [DONE]\n\n"

    def workers(self) -> Dict[str, str]:
        """Stage s is reported as worker_{s+1}. online = ready with a fresh heartbeat; error = the
        stage reported a failure; offline = its process is gone or silent; not_configured = it
        never registered. Same vocabulary as orchestration.py:306-329."""
        import os
        from ..utils.procs import pid_alive
        out = {}
        board = self.pipe.board
        limit_ms = self.cfg.health_timeout_s * 1000
        for s in range(self.cfg.pp_size):
            r = board.read(s)
            if r["state"] == "unregistered":
                st = "not_configured"
            elif r["state"] == "error":
                st = "error"
            elif r["state"] == "stopped":
                st = "offline"
            else:
                alive = True
                if r["pid"] > 0 and r["pid"] != os.getpid():
                    alive = pid_alive(r["pid"]) and r["age_ms"] <= limit_ms
                st = "online" if alive and r["state"] == "ready" else ("offline" if not alive else "error")
            out[f"worker_{s + 1}"] = st
        return out

    def worker_descriptions(self) -> List[str]:
        board = self.pipe.board
        return [f"stage {s}: layers {board.read(s)['layers']} ({board.read(s)['state']})" for s in range(self.cfg.pp_size)]

    def metrics(self) -> str:
        from .metrics import render
        return render(self.engine.snapshot(), self.pipe.board, self.cfg.pp_size)

    def close(self):
        if not self._pipe_closed:
            self._pipe_closed = True
            self.pipe.close()


class HttpRelayBackend:
    """The reference's own topology and loop (orchestration.py:69-228): this process owns the
    tokenizer, embedding, final norm, lm_head and sampler; decoder layers live behind worker
    URLs. With `incremental=True` (default) only the new tokens' rows are sent and the workers keep
    a KV cache per request; with False the whole sequence is re-sent every step, as the reference does."""

    def __init__(self, cfg: EngineConfig, incremental: bool = True):
        import requests
        from ..models.partition import make_worker_plans
        from ..models.stage_ref import RefStage
        from ..parallel.launcher import load_weights, make_tokenizer, resolve_spec
        self.cfg = cfg
        self.spec = resolve_spec(cfg)
        self.model_name = self.spec.name
        self.urls = [u.rstrip("/") for u in cfg.worker_urls]
        edge, _ = make_worker_plans(self.spec, max(1, len(self.urls)), mode=cfg.partition, ranges=cfg.layer_ranges)
        self.edge = RefStage(self.spec, edge, load_weights(cfg, self.spec, edge, "cpu", torch.float32),
                             dtype=torch.float32, max_seq_len=cfg.max_seq_len)
        self.tok = make_tokenizer(cfg, self.spec)
        self.incremental = incremental
        self.session = requests.Session()          # the reference opens a new connection per hop
        self._ids = itertools.count(1)
        self.stats = {"requests": 0, "completed": 0, "failed": 0, "tokens_out": 0, "tokens_in": 0, "hops": 0}

    def _hop(self, url: str, hidden: torch.Tensor, seq_id, ctx_len: int) -> torch.Tensor:
        body = {"hidden_states": hidden.tolist()}
        if self.incremental:
            body["seq_id"], body["ctx_len"] = seq_id, ctx_len
        r = self.session.post(f"{url}/process", json=body, headers=_HEADERS, timeout=self.cfg.hop_timeout_s)
        self.stats["hops"] += 1
        if r.status_code != 200:
            raise RuntimeError(f"Worker at {url} failed: {r.text}")       # orchestration.py:121-122
        return torch.tensor(r.json()["hidden_states"])

    def generate(self, prompt, max_new_tokens, temperature, top_k=None, top_p=None, seed=None):
        start = time.time()
        top_k = self.cfg.top_k if top_k is None else top_k
        top_p = self.cfg.top_p if top_p is None else top_p
        self.stats["requests"] += 1
        if not self.urls or not all(self.urls):
            return {"error": "Workers not configured!", "status": "failed"}        # orchestration.py:84-88
        seq_id = f"r{next(self._ids)}"
        try:
            text = build_prompt(self.tok, prompt, self.cfg.system_prompt, self.cfg.chat_template)
            ids = encode_prompt(self.tok, text)
            self.stats["tokens_in"] += len(ids)
            n = max(1, min(int(max_new_tokens), self.cfg.max_tokens_cap))
            generated: List[int] = []
            eos = getattr(self.tok, "eos_token_id", None)
            stop = {i for i in (eos, *getattr(self.spec, "extra_eos_token_ids", ())) if i is not None}
            cached = 0
            debug = []
            gen = None
            if seed is not None:                      # a seeded request draws from its own generator, reproducibly
                gen = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
            for step in range(n):
                if self.incremental:
                    new = ids[cached:]
                    pos = torch.arange(cached, len(ids))
                else:
                    new, pos = ids, torch.arange(len(ids))
                hidden = self.edge.embed_tokens(torch.tensor(new), pos).unsqueeze(0)      # [1, S, H]
                for url in self.urls:
                    hidden = self._hop(url, hidden, seq_id, cached)
                cached = len(ids) if self.incremental else 0
                logits = self.edge.head_logits(hidden[0, -1:, :])[0]
                if self.cfg.debug_top5 and step < 3:                                      # orchestration.py:172-178
                    debug.append(top_candidates(logits, temperature, top_k, top_p, 5))
                nxt = sample_token(logits, temperature, top_k, top_p, generator=gen)
                if nxt in stop:
                    break
                ids = ids + [nxt]
                generated.append(nxt)
            out_text = self.tok.decode(generated, skip_special_tokens=True)
            elapsed = time.time() - start
            tps = len(generated) / elapsed if elapsed > 0 else 0
            self.stats["completed"] += 1
            self.stats["tokens_out"] += len(generated)
            res = {"prompt": prompt, "response": out_text, "status": "success", "time_taken": f"{elapsed:.2f}s",
                   "tokens_generated": len(generated), "tokens_per_sec": f"{tps:.2f}"}
            if debug:
                res["top5_first_steps"] = debug
            return res
        except Exception as e:
            self.stats["failed"] += 1
            return {"error": f"Error: {e}", "status": "failed"}                          # orchestration.py:220-228
        finally:
            if self.incremental:
                for url in self.urls:
                    try:
                        self.session.post(f"{url}/release", json={"seq_id": seq_id}, headers=_HEADERS, timeout=5)
                    except Exception:
                        pass

    def workers(self) -> Dict[str, str]:
        out = {}
        for i, url in enumerate(self.urls or [""]):
            key = f"worker_{i + 1}"
            if not url:
                out[key] = "not_configured"
                continue
            try:
                r = self.session.get(f"{url}/health", headers=_HEADERS, timeout=self.cfg.health_timeout_s)
                out[key] = "online" if r.status_code == 200 else "error"
            except Exception:
                out[key] = "offline"
        return out

    def worker_descriptions(self) -> List[str]:
        return [f"Worker {i + 1}: {u or 'Not configured'}" for i, u in enumerate(self.urls)]

    def metrics(self) -> str:
        from .metrics import render
        return render(dict(self.stats))

    def close(self):
        self.session.close()


def readiness(backend):
    """(HTTP status, body) of `GET /ready` — an addition for load balancers and orchestration systems:
    200 only when every stage is online and the engine has not failed; `/health` keeps the reference's
    contract (it says "healthy" as long as the process answers, orchestration.py:297-304)."""
    if backend is None:
        return 503, {"ready": False, "reason": "not initialized"}
    try:
        workers = backend.workers()
    except Exception as e:
        return 503, {"ready": False, "reason": f"{type(e).__name__}: {e}"}
    fatal = getattr(getattr(backend, "engine", None), "fatal", None)
    down = {k: v for k, v in workers.items() if v != "online"}
    if fatal or down:
        return 503, {"ready": False, "reason": fatal or "stages not online", "workers": workers}
    return 200, {"ready": True, "workers": workers}


def sampling_extras(data: Dict) -> Dict:
    """Optional per-request sampling fields of `/generate` (additions: the reference fixes top-k 50 and
    top-p 0.9, orchestration.py:353-354): {"top_k": int >= 0 (0 = off), "top_p": 0 < float <= 1,
    "seed": int}. ValueError on anything else."""
    out = {}
    if data.get("top_k") is not None:
        out["top_k"] = int(data["top_k"])
        if out["top_k"] < 0:
            raise ValueError("top_k must be >= 0")
    if data.get("top_p") is not None:
        out["top_p"] = float(data["top_p"])
        if not 0.0 < out["top_p"] <= 1.0:
            raise ValueError("top_p must be in (0, 1]")
    if data.get("seed") is not None:
        out["seed"] = int(data["seed"]) & 0x7FFFFFFFFFFFFFFF
    return out


def create_app(backend):
    """Flask app exposing the orchestrator routes over `backend` (None = not initialised)."""
    from flask import Flask, Response, jsonify, request
    app = Flask("dlia-orchestrator")
    app.config["backend"] = backend

    @app.after_request
    def _cors(resp):                                                   # orchestration.py:232
        resp.headers["Access-Control-Allow-Origin"] = "*"
        resp.headers["Access-Control-Allow-Headers"] = "Content-Type, ngrok-skip-browser-warning"
        resp.headers["Access-Control-Allow-Methods"] = "GET, POST, OPTIONS"
        return resp

    @app.route("/")
    def home():
        return _status_page(app.config["backend"])

    @app.route("/health", methods=["GET"])
    def health():
        b = app.config["backend"]
        return jsonify({"status": "healthy", "role": "orchestrator",
                        "model": b.model_name if b else None, "version": VERSION})

    @app.route("/workers", methods=["GET"])
    def workers():
        b = app.config["backend"]
        if b is None:
            return jsonify({"worker_1": "not_configured"})
        return jsonify(b.workers())

    @app.route("/generate", methods=["POST", "OPTIONS"])
    def generate():
        if request.method == "OPTIONS":
            return "", 204
        b = app.config["backend"]
        if b is None:
            return jsonify({"error": "Orchestrator not initialized"}), 500        # orchestration.py:335-336
        data = request.get_json(silent=True) or {}
        prompt = data.get("prompt", "")
        max_tokens = data.get("max_tokens", b.cfg.default_max_tokens)
        temperature = data.get("temperature", b.cfg.temperature)
        if not prompt:
            return jsonify({"error": "No prompt provided"}), 400                 # orchestration.py:343-344
        try:
            max_tokens = min(int(max_tokens), b.cfg.max_tokens_cap)              # orchestration.py:347 (cap 30 there)
            temperature = float(temperature)
            extra = sampling_extras(data)
        except (TypeError, ValueError) as e:
            return jsonify({"error": f"max_tokens, temperature, top_k, top_p and seed must be numbers ({e})"}), 400
        return jsonify(b.generate(prompt, max_tokens, temperature, **extra))     # HTTP 200 even when status == failed

    @app.route("/ready", methods=["GET"])
    def ready():
        status, body = readiness(app.config["backend"])
        return jsonify(body), status

    @app.route("/metrics", methods=["GET"])
    def metrics():
        b = app.config["backend"]
        return Response(b.metrics() if b else "", mimetype="text/plain; version=0.0.4")

    return app


def _status_page(b) -> str:
    """`GET /`: what is being served and by which stages (the reference shows its worker URLs:
    orchestration.py:236-295). Own markup."""
    stages = "".join(f"<li>{d}</li>" for d in (b.worker_descriptions() if b else ["not initialized"]))
    model = b.model_name if b else "-"
    state = "serving" if b else "starting"
    return f"""<!DOCTYPE html><html><head><meta charset="utf-8"><title>layer-split LLM inference</title>
<style>body{{font:15px/1.55 system-ui,sans-serif;margin:3rem auto;max-width:46rem;color:#17212b;background:#f6f7f9}}
header{{display:flex;justify-content:space-between;align-items:baseline;border-bottom:2px solid #17212b;margin-bottom:1.2rem}}
.state{{font:600 .8rem ui-monospace,monospace;text-transform:uppercase;letter-spacing:.08em;color:#0a7d38}}
dl{{display:grid;grid-template-columns:9rem 1fr;gap:.3rem 1rem}}dt{{font-weight:600}}dd{{margin:0}}
code{{font:.85rem ui-monospace,monospace;background:#e6e9ee;padding:.1rem .3rem;border-radius:3px}}
ul{{margin:.2rem 0;padding-left:1.1rem}}</style></head>
<body><header><h1>Layer-split LLM inference</h1><span class="state">{state}</span></header>
<dl><dt>model</dt><dd>{model}</dd>
<dt>engine</dt><dd>{VERSION}: pipeline stages on MI355X, paged KV cache, continuous batching,
temperature / top-k / top-p sampling, chat formatting</dd>
<dt>stages</dt><dd><ul>{stages}</ul></dd>
<dt>routes</dt><dd><code>GET /health</code> <code>GET /workers</code> <code>POST /generate</code>
<code>GET /metrics</code> <code>GET /ready</code> <code>/v1/*</code></dd></dl></body></html>"""


def create_asgi_app(backend):
    """The same routes as `create_app` on Starlette, for a production ASGI server (uvicorn): requests
    wait on the engine as coroutines, not in one thread each as under the reference's Werkzeug
    development server (orchestration.py:372)."""
    import json

    from starlette.applications import Starlette
    from starlette.responses import HTMLResponse, JSONResponse, PlainTextResponse, Response, StreamingResponse
    from starlette.routing import Route

    state = {"backend": backend}
    cors = {"Access-Control-Allow-Origin": "*",                              # orchestration.py:232
            "Access-Control-Allow-Headers": "Content-Type, ngrok-skip-browser-warning",
            "Access-Control-Allow-Methods": "GET, POST, OPTIONS"}

    def js(obj, status=200):
        return JSONResponse(obj, status_code=status, headers=cors)

    async def home(request):
        return HTMLResponse(_status_page(state["backend"]), headers=cors)

    async def health(request):
        b = state["backend"]
        return js({"status": "healthy", "role": "orchestrator", "model": b.model_name if b else None, "version": VERSION})

    async def workers(request):
        b = state["backend"]
        return js({"worker_1": "not_configured"} if b is None else b.workers())

    async def generate(request):
        if request.method == "OPTIONS":
            return Response(status_code=204, headers=cors)
        b = state["backend"]
        if b is None:
            return js({"error": "Orchestrator not initialized"}, 500)           # orchestration.py:335-336
        try:
            data = json.loads(await request.body() or b"{}")
            if not isinstance(data, dict):
                data = {}
        except ValueError:
            data = {}
        prompt = data.get("prompt", "")
        if not prompt:
            return js({"error": "No prompt provided"}, 400)                      # orchestration.py:343-344
        try:
            max_tokens = min(int(data.get("max_tokens", b.cfg.default_max_tokens)), b.cfg.max_tokens_cap)   # orchestration.py:347
            temperature = float(data.get("temperature", b.cfg.temperature))
            extra = sampling_extras(data)
        except (TypeError, ValueError) as e:
            return js({"error": f"max_tokens, temperature, top_k, top_p and seed must be numbers ({e})"}, 400)
        if data.get("stream"):
            if not hasattr(b, "astream"):
                return js({"error": "streaming needs the pipeline engine behind this API"}, 400)
            return StreamingResponse(b.astream(prompt, max_tokens, temperature, **extra), media_type="text/event-stream",
                                     headers={**cors, "Cache-Control": "no-cache", "X-Accel-Buffering": "no"})
        if hasattr(b, "agenerate"):
            async def client_gone():            # the body has been read: the next message is the disconnect
                while (await request.receive()).get("type") != "http.disconnect":
                    pass
            return js(await b.agenerate(prompt, max_tokens, temperature, client_gone, **extra))   # HTTP 200 even when status == failed
        from starlette.concurrency import run_in_threadpool
        return js(await run_in_threadpool(lambda: b.generate(prompt, max_tokens, temperature, **extra)))

    async def ready(request):
        from starlette.concurrency import run_in_threadpool
        status, body = await run_in_threadpool(readiness, state["backend"])     # the HTTP relay asks its workers over HTTP
        return js(body, status)

    async def metrics(request):
        b = state["backend"]
        return PlainTextResponse(b.metrics() if b else "", media_type="text/plain; version=0.0.4", headers=cors)

    routes = [Route("/", home), Route("/health", health), Route("/workers", workers),
              Route("/generate", generate, methods=["POST", "OPTIONS"]), Route("/metrics", metrics), Route("/ready", ready)]
    from .openai_api import add_routes
    add_routes(routes, state, cors)           # /v1/models, /v1/completions, /v1/chat/completions
    app = Starlette(routes=routes)
    app.state.dlia = state
    return app


def raise_fd_limit():
    """Soft limit of open files up to the hard limit: one socket per waiting client."""
    try:
        import resource
        soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
        if hard == resource.RLIM_INFINITY or soft < hard:
            want = hard if hard != resource.RLIM_INFINITY else max(soft, 1 << 16)
            resource.setrlimit(resource.RLIMIT_NOFILE, (want, hard))
    except (ImportError, ValueError, OSError):
        pass


def serve(cfg: EngineConfig):
    """Blocking: bring the pipeline (or the HTTP relay) up and serve the API on cfg.host:cfg.port."""
    from ..utils.tracing import get_logger
    log = get_logger()
    raise_fd_limit()
    if cfg.resolved_transport() == "http" or cfg.worker_urls:
        backend = HttpRelayBackend(cfg)
        log.info("orchestrator (HTTP relay) for %s; workers: %s", backend.model_name, cfg.worker_urls)
    else:
        import dataclasses
        from ..parallel.launcher import make_pipeline
        if cfg.resolved_device() == "cuda" and not cfg.isolate:
            # serving on GPUs: every stage in its own process, none in this one — a stage hung inside a
            # device call can then always be killed and replaced while this process keeps answering
            cfg = dataclasses.replace(cfg, isolate=True)
        if cfg.resolved_backend() == "hip" and not cfg.decode_buckets and cfg.max_batch >= 128 \
                and os.environ.get("DLIA_AUTO_DECODE_BUCKETS", "1") != "0":
            # a server sized for many sequences still answers a lone request at small-batch latency
            cfg = dataclasses.replace(cfg, decode_buckets=[b for b in (8, 64, 512) if b < cfg.max_batch])
        backend = EngineBackend(make_pipeline(cfg))
        log.info("orchestrator for %s: %d stage(s), backend %s, transport %s", backend.model_name, cfg.pp_size,
                 cfg.resolved_backend(), cfg.resolved_transport())
    import signal

    def on_term(signum, frame):
        raise SystemExit(0)

    # uvicorn shuts down on SIGTERM and then raises the signal again for whoever handled it before:
    # with the default action the process would end right there, before the `finally` below has closed
    # the pipeline (stage processes, shared memory). With this handler it unwinds through it.
    try:
        signal.signal(signal.SIGTERM, on_term)
    except ValueError:                      # not the main thread (a test driving serve() from a thread)
        pass
    try:
        if cfg.server == "flask":
            create_app(backend).run(host=cfg.host, port=cfg.port, debug=False, threaded=True)
        else:
            import uvicorn
            uvicorn.run(create_asgi_app(backend), host=cfg.host, port=cfg.port, log_level="warning",
                        timeout_keep_alive=30, backlog=4096,
                        timeout_graceful_shutdown=int(cfg.shutdown_grace_s))   # then open requests are cancelled
    finally:
        backend.close()
