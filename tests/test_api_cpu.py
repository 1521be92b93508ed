"""API contract of the orchestrator and the HTTP worker (status codes and JSON schemas of the
reference: orchestration.py:236-356, Worker1.py:185-245), the client library, the HTTP relay
topology with real worker processes on localhost, and fault handling."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest
import torch

from distributed_llm_inference_amd.config import EngineConfig
from distributed_llm_inference_amd.parallel.launcher import LocalPipeline, make_pipeline
from distributed_llm_inference_amd.runtime.sampler import SamplingParams
from distributed_llm_inference_amd.serving.orchestrator import EngineBackend, HttpRelayBackend, create_app
from distributed_llm_inference_amd.serving.worker import WorkerService, build_worker_service, create_worker_app

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cfg(**kw):
    base = dict(model="tiny-llama-test", pp_size=1, max_batch=4, num_micro_batches=2, max_seq_len=160,
                prefill_chunk=32, backend="torch", device="cpu", transport="loopback", seed=3)
    base.update(kw)
    return EngineConfig(**base)


@pytest.fixture(scope="module")
def backend():
    b = EngineBackend(LocalPipeline(cfg(pp_size=2, partition="even")))
    yield b
    b.close()


def test_orchestrator_routes(backend):
    c = create_app(backend).test_client()
    r = c.get("/health")
    assert r.status_code == 200
    assert set(r.get_json()) == {"status", "role", "model", "version"}
    assert r.get_json()["status"] == "healthy" and r.get_json()["role"] == "orchestrator"
    r = c.get("/workers")
    assert r.get_json() == {"worker_1": "online", "worker_2": "online"}
    r = c.get("/")
    assert r.status_code == 200 and b"Layer-split LLM inference" in r.data and b"stage 1" in r.data
    r = c.post("/generate", json={"prompt": "Hello", "max_tokens": 6})
    d = r.get_json()
    assert r.status_code == 200 and d["status"] == "success" and d["prompt"] == "Hello"
    assert {"prompt", "response", "status", "time_taken", "tokens_generated", "tokens_per_sec"} <= set(d)
    assert isinstance(d["time_taken"], str) and d["time_taken"].endswith("s")
    assert isinstance(d["tokens_per_sec"], str) and isinstance(d["tokens_generated"], int) and d["tokens_generated"] <= 6
    assert r.headers["Access-Control-Allow-Origin"] == "*"
    assert c.post("/generate", json={"max_tokens": 3}).status_code == 400
    assert c.post("/generate", json={"prompt": ""}).get_json() == {"error": "No prompt provided"}
    # logical failure -> HTTP 200 + error envelope (orchestration.py:220-228,356)
    r = c.post("/generate", json={"prompt": "x" * 4000})
    assert r.status_code == 200 and r.get_json()["status"] == "failed" and r.get_json()["error"].startswith("Error: ")
    # temperature 0 is greedy here (it is a crash in the reference)
    a = c.post("/generate", json={"prompt": "Hi", "max_tokens": 5, "temperature": 0}).get_json()
    b = c.post("/generate", json={"prompt": "Hi", "max_tokens": 5, "temperature": 0}).get_json()
    assert a["status"] == "success" and a["response"] == b["response"]
    m = c.get("/metrics").data.decode()
    assert "dlia_tokens_generated_total" in m and 'dlia_stage_up{stage="1"' in m


def test_asgi_app_serves_the_same_contract(backend):
    """The production front door (Starlette under uvicorn) answers exactly like the Flask app: same
    routes, status codes, JSON schemas, CORS header and HTTP-200 error envelope; and many requests
    wait concurrently as coroutines."""
    from concurrent.futures import ThreadPoolExecutor
    from starlette.testclient import TestClient
    from distributed_llm_inference_amd.serving.orchestrator import create_asgi_app
    with TestClient(create_asgi_app(backend)) as c:
        r = c.get("/health")
        assert r.status_code == 200 and set(r.json()) == {"status", "role", "model", "version"}
        assert c.get("/workers").json() == {"worker_1": "online", "worker_2": "online"}
        r = c.get("/")
        assert r.status_code == 200 and "Layer-split LLM inference" in r.text and "stage 1" in r.text
        r = c.post("/generate", json={"prompt": "Hello", "max_tokens": 6})
        d = r.json()
        assert r.status_code == 200 and d["status"] == "success" and d["prompt"] == "Hello"
        assert {"prompt", "response", "status", "time_taken", "tokens_generated", "tokens_per_sec"} <= set(d)
        assert isinstance(d["time_taken"], str) and d["time_taken"].endswith("s") and isinstance(d["tokens_per_sec"], str)
        assert r.headers["Access-Control-Allow-Origin"] == "*"
        assert c.post("/generate", json={"max_tokens": 3}).status_code == 400
        assert c.post("/generate", json={"prompt": ""}).json() == {"error": "No prompt provided"}
        assert c.post("/generate", json={"prompt": "x", "max_tokens": "many"}).status_code == 400
        assert c.options("/generate").status_code == 204
        r = c.post("/generate", json={"prompt": "x" * 4000})
        assert r.status_code == 200 and r.json()["status"] == "failed" and r.json()["error"].startswith("Error: ")
        a = c.post("/generate", json={"prompt": "Hi", "max_tokens": 5, "temperature": 0}).json()
        flask_answer = create_app(backend).test_client().post("/generate", json={"prompt": "Hi", "max_tokens": 5, "temperature": 0}).get_json()
        assert a["status"] == "success" and a["response"] == flask_answer["response"]
        assert "dlia_tokens_generated_total" in c.get("/metrics").text
        with ThreadPoolExecutor(16) as ex:
            outs = list(ex.map(lambda i: c.post("/generate", json={"prompt": f"q{i}", "max_tokens": 4, "temperature": 0}).json(), range(32)))
        assert all(o["status"] == "success" for o in outs)
    with TestClient(create_asgi_app(None)) as c:
        r = c.post("/generate", json={"prompt": "x"})
        assert r.status_code == 500 and r.json() == {"error": "Orchestrator not initialized"}
        assert c.get("/workers").json() == {"worker_1": "not_configured"}


def test_a_cancelled_request_stops_and_gives_its_row_and_blocks_back(backend):
    """`Engine.cancel`: a queued request leaves the queue, a running one stops receiving tokens at once
    and its KV blocks return to the pool at the next recomposition; the engine keeps serving."""
    eng = backend.engine
    before = eng.stats["cancelled"]
    sp = SamplingParams(temperature=0.0, top_k=1, top_p=1.0, max_new_tokens=140, stop_on_eos=False)
    running = eng.submit([5, 6, 7], sp)
    deadline = time.time() + 60
    while not running.output and time.time() < deadline:
        time.sleep(0.005)
    assert running.output
    others = [eng.submit([9, i + 1], sp) for i in range(12)]     # more requests than rows (2 x 4): the last ones queue
    queued = others[-1]
    eng.cancel(queued)
    eng.cancel(running)
    running.wait(10)
    queued.wait(10)
    assert running.state == "failed" and running.error == "cancelled" and 0 < len(running.output) < 140
    assert queued.state == "failed" and queued.error == "cancelled" and queued.output == []
    n = len(running.output)
    for r in others[:-1]:
        eng.cancel(r)
    eng.cancel(running)                                           # cancelling twice is harmless
    assert eng.wait_idle(30)
    assert eng.alloc.used == 0 and len(running.output) == n
    assert eng.stats["cancelled"] - before == 13
    assert "dlia_requests_cancelled_total" in backend.metrics()
    assert eng.generate_text("still serving", max_new_tokens=4, temperature=0)["status"] == "success"


def test_a_client_that_disconnects_cancels_its_request(backend):
    """The ASGI handler watches for `http.disconnect` while the request decodes and cancels it in the
    engine; a request that outlives `request_timeout_s` is cancelled the same way."""
    import asyncio
    import dataclasses
    from distributed_llm_inference_amd.serving.orchestrator import create_asgi_app
    eng = backend.engine
    app = create_asgi_app(backend)
    body = json.dumps({"prompt": "tell me everything", "max_tokens": 150, "temperature": 0}).encode()

    async def call(disconnect_after):
        calls, sent = {"n": 0}, []

        async def receive():
            calls["n"] += 1
            if calls["n"] == 1:
                return {"type": "http.request", "body": body, "more_body": False}
            await asyncio.sleep(disconnect_after)
            return {"type": "http.disconnect"}

        async def send(msg):
            sent.append(msg)

        scope = {"type": "http", "asgi": {"version": "3.0"}, "http_version": "1.1", "method": "POST", "scheme": "http",
                 "path": "/generate", "raw_path": b"/generate", "query_string": b"", "root_path": "",
                 "headers": [(b"content-type", b"application/json")], "server": ("127.0.0.1", 80), "client": ("127.0.0.1", 9)}
        await app(scope, receive, send)
        return json.loads(b"".join(m.get("body", b"") for m in sent if m["type"] == "http.response.body"))

    before, toks = eng.stats["cancelled"], eng.stats["tokens_out"]
    out = asyncio.run(call(0.05))
    assert out["status"] == "failed" and "client disconnected" in out["error"]
    assert eng.wait_idle(30) and eng.alloc.used == 0              # (the scheduler thread does the cancelling)
    assert eng.stats["cancelled"] == before + 1
    assert eng.stats["tokens_out"] - toks < 150
    # the same request with a patient client runs to its end
    assert asyncio.run(call(600))["status"] == "success"
    # ... and with a server-side deadline shorter than the generation it is cancelled, not left decoding
    backend.cfg = dataclasses.replace(backend.cfg, request_timeout_s=0.05)
    try:
        out = asyncio.run(call(600))
    finally:
        backend.cfg = dataclasses.replace(backend.cfg, request_timeout_s=200.0)
    assert out["status"] == "failed" and "no result within" in out["error"]
    assert eng.wait_idle(30) and eng.alloc.used == 0 and eng.stats["cancelled"] == before + 2


def test_streaming_generate_sends_the_text_as_it_is_decoded(backend):
    """`"stream": true`: server-sent events whose text pieces concatenate to the response of the final
    result event, which is the dictionary a plain `/generate` returns; greedy tokens are the same with
    and without streaming; closing the stream early cancels the request."""
    import asyncio
    from starlette.testclient import TestClient
    from distributed_llm_inference_amd.serving.orchestrator import create_asgi_app
    eng = backend.engine
    with TestClient(create_asgi_app(backend)) as c:
        plain = c.post("/generate", json={"prompt": "Hello there", "max_tokens": 24, "temperature": 0}).json()
        with c.stream("POST", "/generate", json={"prompt": "Hello there", "max_tokens": 24, "temperature": 0, "stream": True}) as r:
            assert r.status_code == 200 and r.headers["content-type"].startswith("text/event-stream")
            events = [l[6:] for l in r.iter_lines() if l.startswith("data: ")]
        assert events[-1] == "[DONE]"
        final = json.loads(events[-2])
        pieces = [json.loads(e) for e in events[:-2]]
        assert final["status"] == "success" and final["response"] == plain["response"]
        assert final["tokens_generated"] == plain["tokens_generated"] == 24
        assert len(pieces) > 1 and "".join(p["text"] for p in pieces) == final["response"]
        assert [p["tokens_generated"] for p in pieces] == sorted(p["tokens_generated"] for p in pieces)
        # a failing request: the error envelope is the only JSON event
        with c.stream("POST", "/generate", json={"prompt": "x" * 4000, "stream": True}) as r:
            events = [l[6:] for l in r.iter_lines() if l.startswith("data: ")]
        assert events[-1] == "[DONE]" and json.loads(events[0])["status"] == "failed" and len(events) == 2

    # a reader that stops after the first piece: the request is cancelled, its blocks come back
    async def first_piece_only():
        agen = backend.astream("tell me a long story", 150, 0.0)
        first = await agen.__anext__()
        await agen.aclose()
        return first

    before = eng.stats["cancelled"]
    first = asyncio.run(first_piece_only())
    assert first.startswith("data: {") and "text" in json.loads(first[6:])
    assert eng.wait_idle(30) and eng.alloc.used == 0 and eng.stats["cancelled"] == before + 1


def test_streamed_text_equals_the_decoded_output_when_the_request_ends_inside_a_character(backend):
    """A request that ends (EOS is not appended, or the cap is hit) while the detokenizer still
    holds back the bytes of an unfinished character: the stream must flush them on its last pass,
    so that streamed and non-streamed text agree, and must never send an empty piece."""
    import asyncio
    from distributed_llm_inference_amd.runtime.engine import Request
    tok = backend.engine.tok
    ids = tok.encode("ok é", add_special_tokens=False)[:-1]          # ends on the first byte of a two-byte character

    async def run():
        req = Request(rid=10 ** 6, prompt_ids=[1], params=SamplingParams(), eos_id=None)
        req.state = "running"
        got = []

        async def feed():
            for i in range(len(ids)):
                req.output.append(ids[i])
                for cb in req.token_callbacks:
                    cb(req)
                await asyncio.sleep(0.01)
            req.state = "finished"                   # EOS arrives: no further token, only the done flag
            req._set_done()

        feeder = asyncio.ensure_future(feed())
        async for text, n in backend.pieces(req):
            got.append((text, n))
        await feeder
        return got

    got = asyncio.run(run())
    assert all(t for t, _ in got)
    assert "".join(t for t, _ in got) == tok.decode(ids, skip_special_tokens=True)
    assert got[-1][1] == len(ids)


def test_client_library_streams(backend):
    """`DistributedLLMClient.stream` against a real uvicorn socket."""
    import threading
    import uvicorn
    from distributed_llm_inference_amd.serving.client import DistributedLLMClient
    from distributed_llm_inference_amd.serving.orchestrator import create_asgi_app
    port = _free_port()
    server = uvicorn.Server(uvicorn.Config(create_asgi_app(backend), host="127.0.0.1", port=port, log_level="warning"))
    t = threading.Thread(target=server.run, daemon=True)
    t.start()
    try:
        _wait_http(f"http://127.0.0.1:{port}/health")
        cl = DistributedLLMClient(f"http://127.0.0.1:{port}", verbose=False, check=False)
        pieces = list(cl.stream("Hello there", max_tokens=12, temperature=0))
        assert cl.last_result["status"] == "success" and "".join(pieces) == cl.last_result["response"]
        assert cl.generate("Hello there", max_tokens=12, temperature=0)["response"] == cl.last_result["response"]
        assert cl.generate_streaming("Hello there", max_tokens=5)["tokens_generated"] <= 5
    finally:
        server.should_exit = True
        t.join(20)


def test_generate_takes_optional_top_k_top_p_and_seed(backend):
    """Additions to the reference's body (it fixes top-k 50 / top-p 0.9 and has no seed): top_k = 1 is
    greedy at any temperature, a seed makes a sampled answer reproducible, bad values are a 400."""
    from starlette.testclient import TestClient
    from distributed_llm_inference_amd.serving.orchestrator import create_asgi_app
    flask = create_app(backend).test_client()
    with TestClient(create_asgi_app(backend)) as c:
        greedy = c.post("/generate", json={"prompt": "Hi", "max_tokens": 12, "temperature": 0}).json()
        k1 = c.post("/generate", json={"prompt": "Hi", "max_tokens": 12, "temperature": 1.5, "top_k": 1}).json()
        assert k1["status"] == "success" and k1["response"] == greedy["response"]
        body = {"prompt": "Hi", "max_tokens": 12, "temperature": 1.5, "top_k": 0, "top_p": 1.0, "seed": 1234}
        a, b = c.post("/generate", json=body).json(), c.post("/generate", json=body).json()
        other = c.post("/generate", json={**body, "seed": 99}).json()
        assert a["status"] == "success" and a["response"] == b["response"] == flask.post("/generate", json=body).get_json()["response"]
        assert other["status"] == "success"
        for bad in ({"top_p": 0}, {"top_p": 1.5}, {"top_k": -1}, {"top_k": "many"}, {"seed": "x"}):
            assert c.post("/generate", json={"prompt": "Hi", **bad}).status_code == 400
            assert flask.post("/generate", json={"prompt": "Hi", **bad}).status_code == 400


def test_openai_style_routes(backend):
    """`/v1/completions` feeds the prompt as it is, `/v1/chat/completions` applies the chat template
    (a single user message gives the prompt, hence the greedy tokens, of `/generate`); usage counts,
    finish reasons, streaming chunks that concatenate to the non-streamed text, error envelopes."""
    from starlette.testclient import TestClient
    from distributed_llm_inference_amd.serving.orchestrator import create_asgi_app
    eng = backend.engine
    with TestClient(create_asgi_app(backend)) as c:
        assert c.get("/v1/models").json()["data"][0]["id"] == backend.model_name
        ref = c.post("/generate", json={"prompt": "Hello", "max_tokens": 10, "temperature": 0}).json()
        r = c.post("/v1/chat/completions", json={"model": "x", "messages": [{"role": "user", "content": "Hello"}],
                                                 "max_tokens": 10, "temperature": 0})
        d = r.json()
        assert r.status_code == 200 and d["object"] == "chat.completion" and d["id"].startswith("chatcmpl-")
        ch = d["choices"][0]
        assert ch["message"] == {"role": "assistant", "content": ref["response"]} and ch["finish_reason"] == "length"
        assert d["usage"]["completion_tokens"] == 10 and d["usage"]["total_tokens"] == d["usage"]["prompt_tokens"] + 10
        # several turns and content parts
        r = c.post("/v1/chat/completions", json={"messages": [
            {"role": "system", "content": "Be brief."}, {"role": "user", "content": "Hi"},
            {"role": "assistant", "content": "Hello!"}, {"role": "user", "content": [{"type": "text", "text": "And now?"}]}],
            "max_tokens": 4, "temperature": 0})
        assert r.status_code == 200 and r.json()["usage"]["completion_tokens"] <= 4

        raw = c.post("/v1/completions", json={"prompt": "Hello", "max_tokens": 10, "temperature": 0}).json()
        assert raw["object"] == "text_completion" and raw["usage"]["prompt_tokens"] == len(eng.tok.encode("Hello"))
        assert raw["usage"]["prompt_tokens"] < d["usage"]["prompt_tokens"]          # no template around it
        one = c.post("/v1/completions", json={"prompt": ["Hello"], "max_tokens": 10, "temperature": 0}).json()
        assert one["choices"][0]["text"] == raw["choices"][0]["text"]
        seeded = [c.post("/v1/completions", json={"prompt": "Hello", "max_tokens": 10, "seed": 7}).json()["choices"][0]["text"]
                  for _ in range(2)]
        assert seeded[0] == seeded[1]

        for path, body in (("/v1/completions", {"prompt": "Hello", "max_tokens": 10, "temperature": 0, "stream": True}),
                           ("/v1/chat/completions", {"messages": [{"role": "user", "content": "Hello"}], "max_tokens": 10,
                                                     "temperature": 0, "stream": True})):
            with c.stream("POST", path, json=body) as r:
                assert r.status_code == 200 and r.headers["content-type"].startswith("text/event-stream")
                events = [l[6:] for l in r.iter_lines() if l.startswith("data: ")]
            assert events[-1] == "[DONE]"
            chunks = [json.loads(e) for e in events[:-1]]
            if path.endswith("chat/completions"):
                assert chunks[0]["choices"][0]["delta"] == {"role": "assistant", "content": ""}
                text = "".join(k["choices"][0]["delta"].get("content", "") for k in chunks)
                assert text == ref["response"] and chunks[0]["object"] == "chat.completion.chunk"
            else:
                assert "".join(k["choices"][0]["text"] for k in chunks) == raw["choices"][0]["text"]
            assert chunks[-1]["choices"][0]["finish_reason"] == "length" and chunks[-1]["usage"]["completion_tokens"] == 10
            assert all(k["choices"][0]["finish_reason"] is None for k in chunks[:-1])

        # stop strings: the text ends before the first hit, streamed or not, and the request stops decoding
        full = c.post("/v1/completions", json={"prompt": "Hello", "max_tokens": 40, "temperature": 0}).json()["choices"][0]["text"]
        assert len(full) >= 6
        stop = full[3:5]
        want = full[:full.find(stop)]
        before = eng.stats["tokens_out"]
        cut = c.post("/v1/completions", json={"prompt": "Hello", "max_tokens": 40, "temperature": 0, "stop": stop}).json()
        assert cut["choices"][0]["text"] == want and cut["choices"][0]["finish_reason"] == "stop"
        assert eng.wait_idle(30) and eng.stats["tokens_out"] - before < 40
        with c.stream("POST", "/v1/completions", json={"prompt": "Hello", "max_tokens": 40, "temperature": 0,
                                                       "stop": ["\x00never", stop], "stream": True}) as r:
            chunks = [json.loads(l[6:]) for l in r.iter_lines() if l.startswith("data: ") and l != "data: [DONE]"]
        assert "".join(k["choices"][0]["text"] for k in chunks) == want and chunks[-1]["choices"][0]["finish_reason"] == "stop"
        assert c.post("/v1/completions", json={"prompt": "x", "stop": ["a", "b", "c", "d", "e"]}).status_code == 400

        for path, body in (("/v1/completions", {}), ("/v1/completions", {"prompt": "x", "n": 2}),
                           ("/v1/completions", {"prompt": "x", "top_p": 0}), ("/v1/chat/completions", {"messages": []}),
                           ("/v1/chat/completions", {"messages": "hi"}), ("/v1/completions", {"prompt": "x", "max_tokens": "many"})):
            r = c.post(path, json=body)
            assert r.status_code == 400 and r.json()["error"]["type"] == "invalid_request_error", (path, body)
        r = c.post("/v1/completions", json={"prompt": "x" * 4000})
        assert r.status_code == 400 and "max_seq_len" in r.json()["error"]["message"]
        assert c.options("/v1/chat/completions").status_code == 204
    with TestClient(create_asgi_app(None)) as c:
        assert c.post("/v1/completions", json={"prompt": "x"}).status_code == 500
        assert c.get("/v1/models").json() == {"object": "list", "data": []}
    assert eng.wait_idle(30) and eng.alloc.used == 0


def test_a_full_queue_refuses_new_requests_at_once():
    """`max_queue`: with every row busy and the queue at its limit a new request fails immediately
    (error envelope on /generate, 429 on the OpenAI-style routes) instead of waiting for its turn."""
    from starlette.testclient import TestClient
    from distributed_llm_inference_amd.serving.orchestrator import create_asgi_app
    b = EngineBackend(LocalPipeline(cfg(max_batch=2, num_micro_batches=1, max_queue=3, max_seq_len=400)))
    try:
        eng = b.engine
        sp = SamplingParams(temperature=0.0, max_new_tokens=300, stop_on_eos=False)
        held = [eng.submit([5, 6, 7], sp) for _ in range(2)]
        deadline = time.time() + 60
        while not all(r.output for r in held) and time.time() < deadline:
            time.sleep(0.005)
        queued = [eng.submit([8, 9], sp) for _ in range(3)]
        time.sleep(0.05)
        refused = eng.submit([8, 9], sp)
        assert refused.done.is_set() and refused.state == "failed" and refused.error.startswith("queue full")
        with TestClient(create_asgi_app(b)) as c:
            r = c.post("/generate", json={"prompt": "Hi", "max_tokens": 4})
            assert r.status_code == 200 and r.json()["status"] == "failed" and "queue full" in r.json()["error"]
            r = c.post("/v1/completions", json={"prompt": "Hi", "max_tokens": 4})
            assert r.status_code == 429 and r.json()["error"]["type"] == "rate_limit_error"
            assert "dlia_requests_rejected_total 3" in c.get("/metrics").text
            for q in held + queued:
                eng.cancel(q)
            assert eng.wait_idle(30)
            assert c.post("/generate", json={"prompt": "Hi", "max_tokens": 4}).json()["status"] == "success"
    finally:
        b.close()


def test_orchestrator_not_initialised():
    c = create_app(None).test_client()
    r = c.post("/generate", json={"prompt": "x"})
    assert r.status_code == 500 and r.get_json() == {"error": "Orchestrator not initialized"}
    assert c.get("/workers").get_json() == {"worker_1": "not_configured"}


def test_token_cap_is_configurable(backend):
    c = create_app(backend).test_client()
    old = backend.cfg.max_tokens_cap
    backend.cfg.max_tokens_cap = 3                     # the reference hard-codes 30 (orchestration.py:347)
    try:
        d = c.post("/generate", json={"prompt": "count", "max_tokens": 50, "temperature": 0}).get_json()
        assert d["tokens_generated"] <= 3
    finally:
        backend.cfg.max_tokens_cap = old


def test_worker_routes_stateless_and_incremental():
    svc = build_worker_service(cfg(), 0, 2)
    c = create_worker_app(svc).test_client()
    h = c.get("/health").get_json()
    assert h == {"status": "healthy", "role": "worker_1", "layers": "0-2", "model": "tiny-llama-test"}
    assert b"Worker 1" in c.get("/").data
    assert c.post("/process", json={}).status_code == 400
    assert c.post("/process", json={"hidden_states": []}).get_json() == {"error": "No hidden states provided"}
    r = c.post("/process", json={"hidden_states": [[[0.0] * 7]]})
    assert r.status_code == 500 and "error" in r.get_json()
    x = torch.randn(1, 6, 128)
    full = c.post("/process", json={"hidden_states": x.tolist()}).get_json()
    assert full["status"] == "success" and full["worker"] == "worker_1"
    want = torch.tensor(full["hidden_states"])
    assert want.shape == (1, 6, 128)
    # stateless: the same request again gives the same answer (nothing was cached)
    again = torch.tensor(c.post("/process", json={"hidden_states": x.tolist()}).get_json()["hidden_states"])
    assert torch.allclose(again, want, atol=1e-6)
    # incremental: 4 tokens, then 2 more with ctx_len=4 == the last two rows of the full pass
    c.post("/process", json={"hidden_states": x[:, :4].tolist(), "seq_id": "a", "ctx_len": 0})
    inc = c.post("/process", json={"hidden_states": x[:, 4:].tolist(), "seq_id": "a", "ctx_len": 4}).get_json()
    assert torch.allclose(torch.tensor(inc["hidden_states"]), want[:, 4:], atol=1e-5)
    assert c.post("/release", json={"seq_id": "a"}).get_json()["released"] == 1
    assert create_worker_app(None).test_client().post("/process", json={"hidden_states": [[[1.0]]]}).status_code == 500


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _wait_http(url, timeout=90):
    import requests
    t0 = time.time()
    while time.time() - t0 < timeout:
        try:
            if requests.get(url, timeout=1).status_code == 200:
                return True
        except Exception:
            time.sleep(0.2)
    return False


@pytest.fixture(scope="module")
def http_workers():
    """Orchestrator + 2 CPU worker PROCESSES on localhost, GPT-2 architecture split across them."""
    ports = [_free_port(), _free_port()]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, "-m", "distributed_llm_inference_amd", "worker", "--model", "tiny-gpt2-test",
                               "--index", str(i), "--num-workers", "2", "--port", str(p), "--host", "127.0.0.1",
                               "--backend", "torch", "--device", "cpu", "--seed", "3", "--max-seq-len", "160"],
                              env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
             for i, p in enumerate(ports)]
    urls = [f"http://127.0.0.1:{p}" for p in ports]
    ok = all(_wait_http(u + "/health") for u in urls)
    yield urls, procs, ok
    for p in procs:
        p.terminate()
    for p in procs:
        p.wait(10)


def test_http_relay_topology_matches_pipeline_engine(http_workers):
    urls, procs, ok = http_workers
    assert ok, "workers did not come up"
    relay = HttpRelayBackend(cfg(model="tiny-gpt2-test", worker_urls=urls, transport="http"))
    stateless = HttpRelayBackend(cfg(model="tiny-gpt2-test", worker_urls=urls, transport="http"), incremental=False)
    eng = EngineBackend(LocalPipeline(cfg(model="tiny-gpt2-test")))
    try:
        assert relay.workers() == {"worker_1": "online", "worker_2": "online"}
        a = relay.generate("Hello there", 8, 0.0)
        b = stateless.generate("Hello there", 8, 0.0)          # whole sequence re-sent every step (the reference)
        c = eng.generate("Hello there", 8, 0.0)
        assert a["status"] == b["status"] == c["status"] == "success"
        assert a["response"] == b["response"] == c["response"]
        assert a["tokens_generated"] == c["tokens_generated"]
        assert stateless.stats["hops"] == relay.stats["hops"]
        # the acceptance client (the reference's Test.py) against a live server
        from distributed_llm_inference_amd.serving.client import DistributedLLMClient
        import threading
        from werkzeug.serving import make_server
        port = _free_port()
        srv = make_server("127.0.0.1", port, create_app(relay), threaded=True)
        t = threading.Thread(target=srv.serve_forever, daemon=True)
        t.start()
        try:
            cl = DistributedLLMClient(f"http://127.0.0.1:{port}/", verbose=False)
            assert cl.api_url == f"http://127.0.0.1:{port}" and cl.check_health() is True
            assert cl.check_workers() == {"worker_1": "online", "worker_2": "online"}
            d = cl.generate("Hello", max_tokens=5)
            assert d and d["status"] == "success"
            said = iter(["health", "workers", "", "Hi", "q"])
            cl.interactive_chat(input_fn=lambda _: next(said), max_tokens=3)
        finally:
            srv.shutdown()
    finally:
        relay.close(); stateless.close(); eng.close()


def test_http_relay_reports_dead_worker(http_workers):
    urls, procs, ok = http_workers
    assert ok
    dead = f"http://127.0.0.1:{_free_port()}"
    relay = HttpRelayBackend(cfg(model="tiny-gpt2-test", worker_urls=[urls[0], dead, ""], transport="http", hop_timeout_s=3))
    try:
        assert relay.workers() == {"worker_1": "online", "worker_2": "offline", "worker_3": "not_configured"}
        out = relay.generate("Hello", 4, 0.7)
        assert out == {"error": "Workers not configured!", "status": "failed"}
        relay.urls = [urls[0], dead]
        out = relay.generate("Hello", 4, 0.7)
        assert out["status"] == "failed" and out["error"].startswith("Error: ")
    finally:
        relay.close()


def test_ready_route(backend):
    from starlette.testclient import TestClient
    from distributed_llm_inference_amd.serving.orchestrator import create_asgi_app
    with TestClient(create_asgi_app(backend)) as c:
        r = c.get("/ready")
        assert r.status_code == 200 and r.json() == {"ready": True, "workers": {"worker_1": "online", "worker_2": "online"}}
    r = create_app(backend).test_client().get("/ready")
    assert r.status_code == 200 and r.get_json()["ready"] is True
    with TestClient(create_asgi_app(None)) as c:
        assert c.get("/ready").status_code == 503
    assert create_app(None).test_client().get("/ready").status_code == 503


def test_stage_failure_fails_requests_and_shows_in_workers():
    """Kill a stage process mid-generation: in-flight requests get the error envelope within the
    timeout and /workers reports the stage offline (parity with orchestration.py:121-122,220-228)."""
    pipe = make_pipeline(cfg(pp_size=2, partition="even", transport="gloo", fault_inject="stage1:exit@4",
                             hop_timeout_s=5))
    b = EngineBackend(pipe)
    try:
        t0 = time.time()
        out = b.generate("Hello", 20, 0.0)
        assert out["status"] == "failed" and out["error"].startswith("Error: ")
        assert time.time() - t0 < 60
        w = b.workers()
        # the killed stage is offline; its neighbour is online — or reports its own error, when its hop saw the
        # connection reset before the scheduler saw the dead process (either order happens)
        assert w["worker_2"] == "offline" and w["worker_1"] in ("online", "error", "offline")
        again = b.generate("Hello", 4, 0.0)               # the engine stays failed, it does not hang
        assert again["status"] == "failed"
        from distributed_llm_inference_amd.serving.orchestrator import readiness
        status, body = readiness(b)                       # ... and a load balancer is told so
        assert status == 503 and body["ready"] is False and body["workers"]["worker_2"] == "offline"
    finally:
        b.close()


def test_auto_restart_rebuilds_the_pipeline_after_a_stage_died():
    """Elastic recovery: with auto_restart the next request after a fatal stage failure gets a
    fresh pipeline (new stage processes, new process groups) and succeeds."""
    pipe = make_pipeline(cfg(pp_size=2, partition="even", transport="gloo", fault_inject="stage1:exit@3",
                             hop_timeout_s=5, auto_restart=True))
    b = EngineBackend(pipe)
    try:
        first = b.generate("Hello", 20, 0.0)
        assert first["status"] == "failed"
        assert b.workers()["worker_2"] == "offline"
        second = b.generate("Hello", 6, 0.0)
        assert second["status"] == "success" and second["tokens_generated"] == 6, second
        assert b.restarts == 1
        assert b.workers() == {"worker_1": "online", "worker_2": "online"}
    finally:
        b.close()


def test_a_failed_bring_up_leaves_no_shared_memory_behind():
    """A stage that raises while coming up (here: the HIP executor asked for on a machine without a
    GPU): the constructor raises, and since the caller gets no object to close, the attempt's rings
    and status board must be gone from /dev/shm already."""
    import glob
    if torch.cuda.device_count() > 0:
        pytest.skip("needs a machine without a GPU to make the stage fail at bring-up")
    before = set(glob.glob(f"/dev/shm/dlia_{os.getpid()}_*"))
    with pytest.raises(RuntimeError, match="bring-up failed"):
        make_pipeline(cfg(pp_size=2, partition="even", transport="gloo", backend="hip", device="cpu"))
    assert set(glob.glob(f"/dev/shm/dlia_{os.getpid()}_*")) == before


def test_a_restart_that_fails_answers_with_the_envelope_and_is_not_retried_by_every_request():
    """auto_restart with a pipeline that cannot be rebuilt: requests get the reference's error
    envelope (never an exception out of the route), and the rebuild is attempted once per back-off
    interval, not once per request."""
    calls = []

    def factory(c):
        calls.append(time.monotonic())
        raise RuntimeError("no GPU left")

    b = EngineBackend(LocalPipeline(cfg(auto_restart=True)), factory=factory)
    try:
        b.engine._fatal("stage 0 failed: injected")
        outs = [b.generate("Hello", 4, 0.0) for _ in range(5)]
        assert all(o["status"] == "failed" and "restart failed" in o["error"] for o in outs), outs
        assert len(calls) == 1 and b.restart_failures == 1 and b.restarts == 0
        # the asynchronous routes answer the same way
        import asyncio
        out = asyncio.run(b.agenerate("Hello", 4, 0.0))
        assert out["status"] == "failed" and len(calls) == 1
        b._next_restart_at = 0.0                       # the back-off has passed
        assert b.generate("Hello", 4, 0.0)["status"] == "failed" and len(calls) == 2
    finally:
        b.close()


def test_a_stage_that_hangs_forever_fails_the_request_in_time_and_is_replaced():
    """A stage stuck inside a call that never returns (here: a sleep of an hour; on a GPU: a receive
    spinning on a dead peer). The serving process holds no stage itself, so its watchdog answers the
    request with the error envelope within hop_timeout_s x stages, /workers shows the pipeline is
    not healthy, and the next request runs on freshly spawned stage processes (the hung one is
    killed, not drained). Parity: orchestration.py:118-122,220-228,306-329."""
    pipe = make_pipeline(cfg(pp_size=2, partition="even", transport="gloo", fault_inject="stage1:stall@3:3600",
                             hop_timeout_s=2, auto_restart=True))
    b = EngineBackend(pipe)
    try:
        old_pids = [p.pid for p in pipe.procs]
        t0 = time.time()
        first = b.generate("Hello", 20, 0.0)
        took = time.time() - t0
        assert first["status"] == "failed" and first["error"].startswith("Error: ")
        assert took < 2 * 2 + 10, took                    # hop_timeout_s x P, plus scheduling slack
        second = b.generate("Hello", 6, 0.0)
        assert second["status"] == "success" and second["tokens_generated"] == 6, second
        assert b.restarts == 1 and b.workers() == {"worker_1": "online", "worker_2": "online"}
        from distributed_llm_inference_amd.utils.procs import pid_alive
        assert not any(pid_alive(p) for p in old_pids)    # the hung stage did not survive the restart
    finally:
        b.close()


def test_single_stage_can_run_in_its_own_process():
    """`isolate`: even one stage lives in a child process; the scheduler's process reaches it over
    the descriptor and result rings only."""
    pipe = make_pipeline(cfg(isolate=True))
    try:
        assert len(pipe.procs) == 1 and pipe.engine.head.remote
        out = pipe.engine.generate_text("Hello", max_new_tokens=5, temperature=0.0)
        assert out["status"] == "success" and out["tokens_generated"] == 5
        ref = LocalPipeline(cfg())
        try:
            assert ref.engine.generate_text("Hello", max_new_tokens=5, temperature=0.0)["response"] == out["response"]
        finally:
            ref.close()
    finally:
        pipe.close()


def test_injected_exception_on_a_thread_stage_is_reported():
    pipe = LocalPipeline(cfg(pp_size=2, partition="even", fault_inject="stage1:raise@2", hop_timeout_s=2))
    try:
        r = pipe.engine.submit([5, 6, 7], SamplingParams(temperature=0.0, max_new_tokens=10, stop_on_eos=False))
        r.wait(60)
        assert r.state == "failed"
        assert pipe.board.read(1)["state"] == "error" and "injected fault" in pipe.board.read(1)["error"]
    finally:
        pipe.close()
