"""Soak of the HTTP front door with impatient clients: a closed loop of clients against a spawned
`serve`, each request at random a plain `/generate`, a streamed one read to its end, a streamed one
dropped after its first events, or a plain one whose client gives up after a random delay. At the
end the server must be idle (nothing running or waiting, no KV block in use), must have cancelled
what was dropped, and must still answer.

    python bench/abort_soak.py --model llama-3-8b --batch 256 --clients 256 --seconds 60
"""
import argparse
import asyncio
import json
import os
import random
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


def tree_rss_mb(pid):
    """Resident memory of the server process and its stage processes."""
    import psutil
    try:
        root = psutil.Process(pid)
        return round(sum(p.memory_info().rss for p in [root] + root.children(recursive=True)) / 2**20, 1)
    except psutil.Error:
        return None


async def soak(url, clients, seconds, max_tokens, seed, server_pid=None):
    import aiohttp
    rng = random.Random(seed)
    stats = {"plain_ok": 0, "stream_ok": 0, "stream_dropped": 0, "gave_up": 0, "failed": 0, "tokens": 0}
    t_end = time.perf_counter() + seconds
    conn = aiohttp.TCPConnector(limit=clients)
    async with aiohttp.ClientSession(connector=conn) as cl:
        async def client(i):
            while time.perf_counter() < t_end:
                kind = rng.choice(["plain", "stream", "drop", "give_up"])
                body = {"prompt": " ".join(f"w{rng.randrange(997)}" for _ in range(rng.randrange(4, 40))),
                        "max_tokens": rng.randrange(8, max_tokens + 1), "temperature": 0.7}
                try:
                    if kind == "plain":
                        async with cl.post(url + "/generate", json=body) as r:
                            d = await r.json()
                            if d.get("status") == "success":
                                stats["plain_ok"] += 1; stats["tokens"] += d["tokens_generated"]
                            else:
                                stats["failed"] += 1
                    elif kind == "give_up":
                        try:
                            async with cl.post(url + "/generate", json=body,
                                               timeout=aiohttp.ClientTimeout(total=rng.uniform(0.02, 0.6))) as r:
                                d = await r.json()
                                stats["plain_ok" if d.get("status") == "success" else "failed"] += 1
                        except asyncio.TimeoutError:
                            stats["gave_up"] += 1
                    else:
                        body["stream"] = True
                        seen, final = 0, None
                        async with cl.post(url + "/generate", json=body) as r:
                            async for line in r.content:
                                if not line.startswith(b"data: ") or line.startswith(b"data: [DONE]"):
                                    continue
                                ev = json.loads(line[6:])
                                seen += 1
                                if "status" in ev:
                                    final = ev
                                elif kind == "drop" and seen >= 2:
                                    break                       # leaving the block closes the connection
                        if final is not None and final.get("status") == "success":
                            stats["stream_ok"] += 1; stats["tokens"] += final["tokens_generated"]
                        elif final is None and kind == "drop":
                            stats["stream_dropped"] += 1
                        else:
                            stats["failed"] += 1
                except (aiohttp.ClientError, asyncio.TimeoutError, ConnectionError):
                    stats["failed"] += 1
        rss = {}

        async def watch_memory():                  # host memory of the server early in the run and at its end
            await asyncio.sleep(min(20.0, seconds * 0.25))
            rss["early"] = tree_rss_mb(server_pid)
            await asyncio.sleep(max(0.0, t_end - time.perf_counter() - 1.0))
            rss["end"] = tree_rss_mb(server_pid)

        async def progress():                      # a line every 30 s: long runs must not look hung
            t0 = time.perf_counter()
            while time.perf_counter() < t_end - 1.0:
                await asyncio.sleep(min(30.0, max(0.1, t_end - time.perf_counter())))
                print(f"[{time.perf_counter() - t0:5.0f} s] {json.dumps(stats)}", file=sys.stderr, flush=True)

        await asyncio.gather(*(client(i) for i in range(clients)), progress(), *([watch_memory()] if server_pid else []))
        stats["server_rss_mb_early"], stats["server_rss_mb_end"] = rss.get("early"), rss.get("end")
        await asyncio.sleep(1.0)
        deadline = time.perf_counter() + 30
        while True:                                             # the scheduler retires cancelled rows at its next recomposition
            async with cl.get(url + "/metrics") as r:
                m = {l.split()[0]: float(l.split()[1]) for l in (await r.text()).splitlines() if l and not l.startswith("#") and "{" not in l}
            idle = m.get("dlia_requests_running") == 0 and m.get("dlia_requests_waiting") == 0 and m.get("dlia_kv_blocks_used") == 0
            if idle or time.perf_counter() > deadline:
                break
            await asyncio.sleep(0.2)
        async with cl.post(url + "/generate", json={"prompt": "still there?", "max_tokens": 8}) as r:
            after = (await r.json()).get("status")
    return stats, m, idle, after


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--device", default="auto")
    ap.add_argument("--pp-size", type=int, default=1)
    ap.add_argument("--transport", default="auto")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--micro-batches", type=int, default=2)
    ap.add_argument("--max-seq-len", type=int, default=512)
    ap.add_argument("--clients", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--max-tokens", type=int, default=96)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    port = free_port()
    url = f"http://127.0.0.1:{port}"
    cmd = [sys.executable, "-m", "distributed_llm_inference_amd", "serve", "--model", a.model, "--dtype", a.dtype,
           "--device", a.device, "--pp-size", str(a.pp_size), "--transport", a.transport, "--max-batch", str(a.batch),
           "--num-micro-batches", str(a.micro_batches), "--max-seq-len", str(a.max_seq_len),
           "--max-tokens-cap", str(a.max_tokens), "--host", "127.0.0.1", "--port", str(port)]
    proc = subprocess.Popen(cmd, cwd=ROOT, start_new_session=True)
    try:
        import httpx
        t0 = time.time()
        while True:
            try:
                if httpx.get(url + "/health", timeout=2).status_code == 200:
                    break
            except Exception:
                pass
            if proc.poll() is not None or time.time() - t0 > 600:
                raise SystemExit("the server did not come up")
            time.sleep(0.5)
        stats, m, idle, after = asyncio.run(soak(url, a.clients, a.seconds, a.max_tokens, a.seed, proc.pid))
        out = {"seconds": a.seconds, "clients": a.clients, **stats,
               "server": {k[5:]: m.get(k) for k in ("dlia_requests_total", "dlia_requests_completed_total", "dlia_requests_cancelled_total",
                                                    "dlia_requests_failed_total", "dlia_tokens_generated_total", "dlia_requests_running",
                                                    "dlia_requests_waiting", "dlia_kv_blocks_used", "dlia_preemptions_total")},
               "idle_at_the_end": idle, "answers_afterwards": after, "model": a.model, "pp_size": a.pp_size,
               "micro_batch": a.batch, "micro_batches": a.micro_batches}
        print(json.dumps(out))
        ok = idle and after == "success" and stats["failed"] == 0 and m.get("dlia_requests_cancelled_total", 0) > 0
        sys.exit(0 if ok else 1)
    finally:
        proc.send_signal(signal.SIGTERM)
        try:
            proc.wait(60)
        except subprocess.TimeoutExpired:
            pass
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass


if __name__ == "__main__":
    main()
