"""Command line and configuration plumbing: YAML / environment / flag precedence, `plan`,
`generate`, and the `serve` process answering the reference's API over a real socket."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

from distributed_llm_inference_amd.config import EngineConfig
from distributed_llm_inference_amd.serving import cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(argv):
    import argparse
    ap = argparse.ArgumentParser()
    cli._add_engine_args(ap)
    return ap.parse_args(argv)


def test_config_precedence_yaml_env_flags(tmp_path, monkeypatch):
    y = tmp_path / "engine.yaml"
    y.write_text("model: tiny-llama-test\npp_size: 2\nmax_batch: 8\ntop_k: 40\nlayer_ranges: [[0, 1], [1, 4]]\n")
    monkeypatch.setenv("DLIA_MAX_BATCH", "16")
    monkeypatch.setenv("DLIA_TEMPERATURE", "0.5")
    cfg = cli.config_from_args(_args(["--config", str(y), "--max-seq-len", "64", "--trace", "/tmp/t.json", "--dtype", "fp8"]))
    assert cfg.model == "tiny-llama-test" and cfg.pp_size == 2          # YAML
    assert cfg.max_batch == 16 and cfg.temperature == 0.5               # environment over YAML
    assert cfg.max_seq_len == 64 and cfg.trace_path == "/tmp/t.json"    # flags over both
    assert cfg.dtype == "fp8" and cfg.top_k == 40
    assert cfg.partition == "explicit" and cfg.layer_ranges == [(0, 1), (1, 4)]
    with pytest.raises(ValueError):
        EngineConfig.from_dict({"no_such_field": 1})
    cfg = cli.config_from_args(_args(["--max-batch", "256", "--decode-buckets", "64, 8,512", "--max-queue", "100", "--request-timeout-s", "30"]))
    assert cfg.decode_buckets == [64, 8, 512] and cfg.graph_sizes() == [8, 64, 256]      # sizes at or above max_batch drop out
    assert cfg.max_queue == 100 and cfg.request_timeout_s == 30.0
    assert EngineConfig(max_batch=32).graph_sizes() == [32]


def test_plan_command_prints_the_reference_split(capsys):
    assert cli.main(["plan", "--model", "tinyllama-1.1b", "--pp-size", "2", "--partition", "even"]) == 0
    out = capsys.readouterr().out
    assert "stage 0: layers 0-11 + embedding" in out          # Worker1.py:27-28
    assert "stage 1: layers 11-22 + final norm + lm_head + sampler" in out   # Worker2.py:26-27
    cli.main(["plan", "--model", "llama-3-70b", "--pp-size", "8"])
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == 8 and lines[-1].startswith("stage 7: layers ")


def test_generate_command(capsys):
    rc = cli.main(["generate", "Hello", "--model", "tiny-llama-test", "--device", "cpu", "--backend", "torch",
                   "--max-tokens", "5", "--temperature", "0", "--max-seq-len", "128"])
    assert rc == 0
    d = json.loads(capsys.readouterr().out)
    assert d["status"] == "success" and d["tokens_generated"] <= 5 and d["prompt"] == "Hello"


def test_serve_process_speaks_the_reference_api():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.Popen([sys.executable, "-m", "distributed_llm_inference_amd", "serve", "--model", "tiny-llama-test",
                          "--device", "cpu", "--backend", "torch", "--pp-size", "2", "--transport", "loopback",
                          "--host", "127.0.0.1", "--port", str(port), "--max-seq-len", "160"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        import requests
        from distributed_llm_inference_amd.serving.client import DistributedLLMClient
        url = f"http://127.0.0.1:{port}"
        t0 = time.time()
        while True:
            try:
                if requests.get(url + "/health", timeout=2).status_code == 200:
                    break
            except Exception:
                assert p.poll() is None, p.stdout.read()
                assert time.time() - t0 < 120
                time.sleep(0.3)
        c = DistributedLLMClient(url)
        assert c.check_workers() == {"worker_1": "online", "worker_2": "online"}
        r = c.generate("Hello", max_tokens=4)
        assert r is not None and r["status"] == "success"
    finally:
        p.terminate()
        try:
            p.wait(20)
        except Exception:
            p.kill()
