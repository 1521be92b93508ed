"""OpenAI-style routes next to the reference's own (`GET /v1/models`, `POST /v1/completions`,
`POST /v1/chat/completions`), for clients that already speak that protocol. An addition: the
reference's API is the four routes of orchestration.py:230-356, which `orchestrator.py` keeps.

Supported request fields: `prompt` (a string, or a list holding one string) / `messages`,
`max_tokens`, `temperature`, `top_p`, `top_k` (extension), `seed`, `stream`, `stop`, `n` (1 only). Other
fields are ignored. `/v1/completions` feeds the prompt to the model as it is; `/v1/chat/completions`
formats the messages with the tokenizer's chat template (the reference's Zephyr layout without one).
Errors use the `{"error": {"message", "type", "code"}}` envelope with a 4xx / 5xx status.
"""
from __future__ import annotations

import asyncio
import contextlib
import json
import time
from typing import Dict, List, Optional

from ..runtime.sampler import SamplingParams
from ..runtime.tokenizer import build_chat_prompt, encode_prompt


class BadRequest(ValueError):
    pass


def _error(message: str, status: int = 400, typ: str = "invalid_request_error") -> Dict:
    return {"error": {"message": message, "type": typ, "param": None, "code": status}}


def parse_sampling(data: Dict, cfg) -> SamplingParams:
    """OpenAI defaults (temperature 1, top_p 1, no top-k, 16 tokens) where the body is silent."""
    try:
        n = int(data.get("n", 1) or 1)
        if n != 1:
            raise BadRequest("only n = 1 is supported")
        max_tokens = data.get("max_tokens", data.get("max_completion_tokens"))
        max_tokens = 16 if max_tokens is None else int(max_tokens)
        if max_tokens < 1:
            raise BadRequest("max_tokens must be at least 1")
        temperature = 1.0 if data.get("temperature") is None else float(data["temperature"])
        top_p = 1.0 if data.get("top_p") is None else float(data["top_p"])
        top_k = 0 if data.get("top_k") is None else int(data["top_k"])
        seed = None if data.get("seed") is None else int(data["seed"]) & 0x7FFFFFFFFFFFFFFF
    except (TypeError, ValueError) as e:
        if isinstance(e, BadRequest):
            raise
        raise BadRequest(f"max_tokens, temperature, top_p, top_k, seed and n must be numbers ({e})")
    if temperature < 0 or not 0.0 < top_p <= 1.0 or top_k < 0:
        raise BadRequest("temperature must be >= 0, top_p in (0, 1], top_k >= 0")
    return SamplingParams(temperature=temperature, top_k=top_k, top_p=top_p,
                          max_new_tokens=min(max_tokens, cfg.max_tokens_cap), seed=seed)


def parse_stop(data: Dict) -> List[str]:
    stop = data.get("stop")
    if stop is None:
        return []
    if isinstance(stop, str):
        stop = [stop]
    if not isinstance(stop, list) or len(stop) > 4 or not all(isinstance(x, str) and x for x in stop):
        raise BadRequest("stop must be a non-empty string or a list of up to 4 of them")
    return stop


class StopFilter:
    """Cuts streamed text at the first occurrence of any stop string. Text that could still turn out
    to be the beginning of a stop string (the last `longest - 1` characters) is held back."""

    def __init__(self, stops: List[str]):
        self.stops, self.buf, self.hit = stops, "", False
        self.hold = max((len(x) for x in stops), default=1) - 1

    def feed(self, piece: str) -> str:
        if self.hit:
            return ""
        self.buf += piece
        cut = min((i for i in (self.buf.find(x) for x in self.stops) if i >= 0), default=-1)
        if cut >= 0:
            out, self.buf, self.hit = self.buf[:cut], "", True
            return out
        keep = min(self.hold, len(self.buf))
        out, self.buf = self.buf[:len(self.buf) - keep], self.buf[len(self.buf) - keep:]
        return out

    def flush(self) -> str:
        out, self.buf = ("" if self.hit else self.buf), ""
        return out


def _finish_reason(req, cfg) -> str:
    hit_cap = len(req.output) >= req.params.max_new_tokens or len(req.prompt_ids) + len(req.output) >= cfg.max_seq_len
    return "length" if hit_cap else "stop"


def _usage(req) -> Dict:
    return {"prompt_tokens": len(req.prompt_ids), "completion_tokens": len(req.output),
            "total_tokens": len(req.prompt_ids) + len(req.output)}


def add_routes(routes: List, state: Dict, cors: Dict):
    """Appends the three routes to a Starlette route list; `state["backend"]` is the orchestrator's backend."""
    from starlette.responses import JSONResponse, Response, StreamingResponse
    from starlette.routing import Route

    def js(obj, status=200):
        return JSONResponse(obj, status_code=status, headers=cors)

    def backend_or_error():
        b = state["backend"]
        if b is None:
            return None, js(_error("Orchestrator not initialized", 500, "server_error"), 500)
        if not hasattr(b, "wait_request"):
            return None, js(_error("these routes need the pipeline engine behind the API", 501, "server_error"), 501)
        return b, None

    async def models(request):
        b = state["backend"]
        data = [] if b is None else [{"id": b.model_name, "object": "model", "created": 0, "owned_by": "dlia"}]
        return js({"object": "list", "data": data})

    async def body_of(request) -> Dict:
        try:
            data = json.loads(await request.body() or b"{}")
        except ValueError:
            raise BadRequest("the body is not JSON")
        if not isinstance(data, dict):
            raise BadRequest("the body must be a JSON object")
        return data

    async def run(request, b, kind: str, prompt_ids: List[int], sp: SamplingParams, stream: bool, stops: List[str]):
        """Common tail of both POST routes. kind: "text_completion" | "chat.completion"."""
        await b.recover_if_needed()
        engine = b.engine
        created = int(time.time())
        req = engine.submit(prompt_ids, sp)
        rid = ("cmpl-" if kind == "text_completion" else "chatcmpl-") + str(req.rid)

        def choice(text: Optional[str], finish: Optional[str], first: bool = False) -> Dict:
            if kind == "text_completion":
                return {"index": 0, "text": text or "", "logprobs": None, "finish_reason": finish}
            if not stream:
                return {"index": 0, "message": {"role": "assistant", "content": text or ""}, "logprobs": None,
                        "finish_reason": finish}
            delta = {"role": "assistant", "content": ""} if first else ({"content": text} if text else {})
            return {"index": 0, "delta": delta, "logprobs": None, "finish_reason": finish}

        def failed():
            return _error(req.error or "generation failed", 500, "server_error")

        async def filtered():
            """(text, done-by-stop) pieces with the stop strings applied; the request is cancelled at a hit."""
            f = StopFilter(stops)
            async with contextlib.aclosing(b.pieces(req)) as pieces:
                async for text, _n in pieces:
                    out = f.feed(text) if stops else text
                    if out:
                        yield out
                    if f.hit:
                        return                  # leaving `pieces` open-ended closes it, which cancels the request
            tail = f.flush()
            if tail:
                yield tail

        def finish_reason():
            return "stop" if req.error == "cancelled" or not req.finished else _finish_reason(req, b.cfg)

        def failure_response():
            """A failed request as an HTTP answer: back-pressure is 429, a request the engine can
            never serve is 400, anything else 500 — the same on every non-streamed path."""
            err = req.error or ""
            if err.startswith("queue full"):
                return js(_error(err, 429, "rate_limit_error"), 429)
            return js(failed(), 400 if "exceeds max_seq_len" in err or "empty prompt" in err else 500)

        async def client_gone():
            while (await request.receive()).get("type") != "http.disconnect":
                pass

        if not stream and stops:                # the text has to be watched while it grows
            async def collect():
                return "".join([t async for t in filtered()])
            # raced against the client's disconnect: a cancelled collector closes `pieces`, which
            # cancels the request in the engine instead of decoding on for nobody
            job, gone = asyncio.ensure_future(collect()), asyncio.ensure_future(client_gone())
            try:
                await asyncio.wait({job, gone}, return_when=asyncio.FIRST_COMPLETED)
                if not job.done():
                    job.cancel()
                    with contextlib.suppress(asyncio.CancelledError, Exception):
                        await job
                    return js(_error("client disconnected", 408, "timeout"), 408)
                text = job.result()
            except TimeoutError as e:
                return js(_error(str(e), 408, "timeout"), 408)
            finally:
                gone.cancel()
            if req.state == "failed" and req.error != "cancelled":
                return failure_response()
            return js({"id": rid, "object": kind, "created": created, "model": b.model_name,
                       "choices": [choice(text, finish_reason())], "usage": _usage(req)})

        if not stream:
            why = await b.wait_request(req, client_gone)
            if why is not None:
                return js(_error(why, 408, "timeout"), 408)
            if req.state == "failed":
                return failure_response()
            text = engine.tok.decode(req.output, skip_special_tokens=True)
            return js({"id": rid, "object": kind, "created": created, "model": b.model_name,
                       "choices": [choice(text, _finish_reason(req, b.cfg))], "usage": _usage(req)})

        obj = kind if kind == "text_completion" else "chat.completion.chunk"

        def chunk(c: Dict, usage: Optional[Dict] = None) -> str:
            d = {"id": rid, "object": obj, "created": created, "model": b.model_name, "choices": [c]}
            if usage is not None:
                d["usage"] = usage
            return f"data: {json.dumps(d)}\n\n"

        async def events():
            try:
                if kind != "text_completion":
                    yield chunk(choice(None, None, first=True))
                async with contextlib.aclosing(filtered()) as texts:
                    async for text in texts:
                        yield chunk(choice(text, None))
                if req.state == "failed" and req.error != "cancelled":
                    yield f"data: {json.dumps(failed())}\n\n"
                else:
                    yield chunk(choice(None, finish_reason()), _usage(req))
            except Exception as e:
                yield f"data: {json.dumps(_error(str(e) or type(e).__name__, 500, 'server_error'))}\n\n"
            yield "data: [DONE]\n\n"

        return StreamingResponse(events(), media_type="text/event-stream",
                                 headers={**cors, "Cache-Control": "no-cache", "X-Accel-Buffering": "no"})

    async def completions(request):
        if request.method == "OPTIONS":
            return Response(status_code=204, headers=cors)
        b, err = backend_or_error()
        if err is not None:
            return err
        try:
            data = await body_of(request)
            prompt = data.get("prompt")
            if isinstance(prompt, list) and len(prompt) == 1:
                prompt = prompt[0]
            if not isinstance(prompt, str) or not prompt:
                raise BadRequest("prompt must be a non-empty string (or a list holding one)")
            sp = parse_sampling(data, b.cfg)
            stops = parse_stop(data)
        except BadRequest as e:
            return js(_error(str(e)), 400)
        return await run(request, b, "text_completion", encode_prompt(b.engine.tok, prompt), sp, bool(data.get("stream")), stops)

    async def chat_completions(request):
        if request.method == "OPTIONS":
            return Response(status_code=204, headers=cors)
        b, err = backend_or_error()
        if err is not None:
            return err
        try:
            data = await body_of(request)
            messages = data.get("messages")
            if not isinstance(messages, list) or not messages or not all(isinstance(m, dict) for m in messages):
                raise BadRequest("messages must be a non-empty list of {role, content} objects")
            for m in messages:                 # content parts ([{type: text, text: ...}]) are joined
                if isinstance(m.get("content"), list):
                    m["content"] = "".join(p.get("text", "") for p in m["content"] if isinstance(p, dict))
            sp = parse_sampling(data, b.cfg)
            stops = parse_stop(data)
        except BadRequest as e:
            return js(_error(str(e)), 400)
        text = build_chat_prompt(b.engine.tok, messages, b.cfg.system_prompt, b.cfg.chat_template)
        return await run(request, b, "chat.completion", encode_prompt(b.engine.tok, text), sp, bool(data.get("stream")), stops)

    routes += [Route("/v1/models", models),
               Route("/v1/completions", completions, methods=["POST", "OPTIONS"]),
               Route("/v1/chat/completions", chat_completions, methods=["POST", "OPTIONS"])]
