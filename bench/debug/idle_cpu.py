"""CPU time a spawned `serve` (and its stage processes) burns while idle."""
import os, signal, socket, subprocess, sys, time
import httpx, psutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
pp = sys.argv[1] if len(sys.argv) > 1 else "1"
cmd = [sys.executable, "-m", "distributed_llm_inference_amd", "serve", "--model", "llama-3-8b", "--max-batch", "256", "--pp-size", pp,
       "--max-seq-len", "512", "--host", "127.0.0.1", "--port", str(port)] + (["--transport", "rccl"] if pp != "1" else [])
proc = subprocess.Popen(cmd, cwd=ROOT, start_new_session=True)
try:
    t0 = time.time()
    while True:
        try:
            if httpx.get(f"http://127.0.0.1:{port}/ready", timeout=2).status_code == 200:
                break
        except Exception:
            pass
        if time.time() - t0 > 300 or proc.poll() is not None:
            raise SystemExit("server did not come up")
        time.sleep(0.5)
    httpx.post(f"http://127.0.0.1:{port}/generate", json={"prompt": "hi", "max_tokens": 8}, timeout=60)
    time.sleep(2)
    tree = lambda: [psutil.Process(proc.pid)] + psutil.Process(proc.pid).children(recursive=True)
    cpu = lambda: {p.pid: sum(p.cpu_times()[:2]) for p in tree()}
    a = cpu(); time.sleep(10); b = cpu()
    print({"pp": pp, "idle_cpu_percent_per_process": {pid: round((b[pid] - a[pid]) * 10, 1) for pid in a if pid in b}})
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
