#!/usr/bin/env python3
"""Load test of the HTTP front door: N concurrent clients POST /generate to a running orchestrator
(`python -m distributed_llm_inference_amd serve ...`) or to one this script starts, and the script
reports requests/s, output tokens/s and p50 / p99 request latency — the path the reference serves
from Flask's development server (orchestration.py:331-372), measured end to end through HTTP.

    python bench/http_bench.py --spawn --model llama-3-8b --clients 256 --requests 2048 --max-tokens 32
    python bench/http_bench.py --url http://127.0.0.1:5000 --clients 64 --requests 512
"""
import argparse
import asyncio
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


async def run_load(url, clients, requests, max_tokens, prompt_words, timeout, stream=False):
    """`clients` connections, each posting its share of `requests` one after the other (closed loop).
    aiohttp: with hundreds of open connections the client must not be what is measured."""
    import aiohttp
    lat, toks, fails = [], 0, 0
    ttft = []
    todo = iter(range(requests))
    conn = aiohttp.TCPConnector(limit=clients, force_close=False)
    async with aiohttp.ClientSession(connector=conn, timeout=aiohttp.ClientTimeout(total=timeout)) as cl:
        async def client():
            nonlocal toks, fails
            for i in todo:
                body = {"prompt": " ".join(f"w{(i * 7 + k) % 997}" for k in range(prompt_words)), "max_tokens": max_tokens,
                        "temperature": 0.7}
                t0 = time.perf_counter()
                try:
                    if stream:                 # server-sent events: time to the first piece of text, then the result event
                        body["stream"] = True
                        d, first = {}, None
                        async with cl.post(url + "/generate", json=body) as r:
                            async for line in r.content:
                                if not line.startswith(b"data: ") or line.startswith(b"data: [DONE]"):
                                    continue
                                ev = json.loads(line[6:])
                                if first is None:
                                    first = time.perf_counter() - t0
                                if "status" in ev:
                                    d = ev
                            if r.status == 200 and d.get("status") == "success":
                                toks += int(d["tokens_generated"])
                                lat.append(time.perf_counter() - t0)
                                ttft.append(first)
                            else:
                                fails += 1
                        continue
                    async with cl.post(url + "/generate", json=body) as r:
                        d = await r.json()
                        if r.status == 200 and d.get("status") == "success":
                            toks += int(d["tokens_generated"])
                            lat.append(time.perf_counter() - t0)
                        else:
                            fails += 1
                except Exception:
                    fails += 1
        t0 = time.perf_counter()
        await asyncio.gather(*(client() for _ in range(clients)))
        wall = time.perf_counter() - t0
    lat.sort()
    ttft.sort()
    pick = lambda q: round(lat[min(len(lat) - 1, int(q * len(lat)))] * 1e3, 1) if lat else None
    extra = {}
    if ttft:
        extra = {"stream": True, "client_ttft_ms_p50": round(ttft[len(ttft) // 2] * 1e3, 1),
                 "client_ttft_ms_p99": round(ttft[min(len(ttft) - 1, int(0.99 * len(ttft)))] * 1e3, 1)}
    return {**extra, "requests": requests, "clients": clients, "failed": fails, "wall_s": round(wall, 3),
            "requests_per_s": round((requests - fails) / wall, 2), "output_tokens_per_s": round(toks / wall, 1),
            "latency_ms_p50": pick(0.5), "latency_ms_p99": pick(0.99), "max_tokens": max_tokens}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--url")
    ap.add_argument("--spawn", action="store_true", help="start an orchestrator on a free port for the run")
    ap.add_argument("--server", default="uvicorn", choices=["uvicorn", "flask"])
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--device", default="auto")
    ap.add_argument("--pp-size", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--micro-batches", type=int, default=2)
    ap.add_argument("--max-seq-len", type=int, default=1024)
    ap.add_argument("--clients", type=int, default=256)
    ap.add_argument("--requests", type=int, default=1024)
    ap.add_argument("--max-tokens", type=int, default=32)
    ap.add_argument("--prompt-words", type=int, default=16)
    ap.add_argument("--timeout", type=float, default=600.0)
    ap.add_argument("--transport", default="auto", help="stage transport of the spawned orchestrator (gloo: stages may share one GPU)")
    ap.add_argument("--kv-dtype", default="bf16")
    ap.add_argument("--decode-buckets", default="", help='passed to `serve --decode-buckets`, e.g. "8,64"')
    ap.add_argument("--stream", action="store_true", help="clients read server-sent events; reports the time to the first piece of text")
    ap.add_argument("--show-one", action="store_true", help="first print the whole JSON answer of one /generate")
    a = ap.parse_args()
    import resource                      # one socket per client
    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
    if hard != resource.RLIM_INFINITY and soft < hard:
        resource.setrlimit(resource.RLIMIT_NOFILE, (hard, hard))
    proc = None
    url = a.url
    if a.spawn or not url:
        port = free_port()
        url = f"http://127.0.0.1:{port}"
        cmd = [sys.executable, "-m", "distributed_llm_inference_amd", "serve", "--model", a.model, "--dtype", a.dtype,
               "--device", a.device, "--pp-size", str(a.pp_size), "--max-batch", str(a.batch),
               "--num-micro-batches", str(a.micro_batches), "--max-seq-len", str(a.max_seq_len),
               "--max-tokens-cap", str(max(a.max_tokens, 32)), "--host", "127.0.0.1", "--port", str(port),
               "--server", a.server, "--transport", a.transport, "--kv-dtype", a.kv_dtype]
        if a.decode_buckets:
            cmd += ["--decode-buckets", a.decode_buckets]
        t_spawn = time.time()
        proc = subprocess.Popen(cmd, cwd=ROOT, start_new_session=True)
        import httpx
        t0 = time.time()
        while True:
            try:
                if httpx.get(url + "/health", timeout=2).status_code == 200:
                    break
            except Exception:
                pass
            if proc.poll() is not None or time.time() - t0 > 900:
                raise SystemExit("orchestrator did not come up")
            time.sleep(0.5)
    try:
        if proc is not None:
            print(json.dumps({"startup_s": round(time.time() - t_spawn, 1)}), flush=True)
        if a.show_one:
            import httpx
            t0 = time.perf_counter()
            r = httpx.post(url + "/generate", json={"prompt": "Explain pipeline parallel inference in one sentence.",
                                                    "max_tokens": a.max_tokens, "temperature": 0.7}, timeout=a.timeout)
            print(json.dumps({"one_generate": r.json(), "http_status": r.status_code,
                              "client_wall_s": round(time.perf_counter() - t0, 3),
                              "workers": httpx.get(url + "/workers", timeout=30).json()}), flush=True)
        asyncio.run(run_load(url, min(a.clients, 8), min(a.requests, 16), a.max_tokens, a.prompt_words, a.timeout))   # warm-up
        out = asyncio.run(run_load(url, a.clients, a.requests, a.max_tokens, a.prompt_words, a.timeout, stream=a.stream))
        out.update({"server": a.server if proc else "external", "model": a.model, "dtype": a.dtype, "pp_size": a.pp_size, "transport": a.transport,
                    "micro_batch": a.batch, "micro_batches": a.micro_batches})
        print(json.dumps(out))
    finally:
        if proc is not None:
            proc.send_signal(signal.SIGTERM)       # the server alone: it closes its stage processes itself
            try:
                proc.wait(60)
            except subprocess.TimeoutExpired:
                pass
            try:
                os.killpg(proc.pid, signal.SIGKILL)    # whatever of its session is still there
            except ProcessLookupError:
                pass


if __name__ == "__main__":
    main()
