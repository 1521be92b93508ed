#!/usr/bin/env python3
"""Baseline: the REFERENCE's algorithm, run locally, on the benchmark models.

The reference publishes no numbers, so the number to beat is its own path on the same model:
Hugging Face `transformers` decoder layers on CPU in fp32, an orchestrator that embeds the WHOLE
sequence every step and relays it through two layer workers as JSON nested lists over HTTP
(no KV cache), final norm + lm_head over all positions, top-k/top-p sampling — i.e. the loop of
orchestration.py:109-196 and the worker of Worker1.py:82-177/208-245, minus ngrok (localhost
instead) and with random-init weights of the named architecture (no checkpoints offline).

Each worker process builds only its own layers (the reference loads the full model in every
process, Worker1.py:60-75 — that would not fit for the larger models and does not change speed).

    python bench/reference_baseline.py --model tinyllama-1.1b --tokens 8
    python bench/reference_baseline.py --model llama-3-8b --tokens 4

Prints one JSON line and updates bench/baseline_numbers.json.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PROMPT = "<|system|>\nYou are a helpful assistant.</s>\n<|user|>\nHello</s>\n<|assistant|>\n"


def hf_config(spec, num_layers):
    import transformers
    d = {k: v for k, v in spec.hf_config().items() if k not in ("model_type", "architectures")}
    if spec.is_llama:
        d["num_hidden_layers"] = num_layers
        return transformers.LlamaConfig(**d)
    d["n_layer"] = num_layers
    return transformers.GPT2Config(**d)


def worker_main(model: str, start: int, end: int, port: int, name: str):
    """A reference-style worker: HF layers [start, end) behind POST /process."""
    import transformers
    from flask import Flask, jsonify, request
    from distributed_llm_inference_amd.models.spec import ModelSpec
    spec = ModelSpec.from_pretrained(model)
    torch.manual_seed(start)
    cfg = hf_config(spec, end - start)
    cfg._attn_implementation = "sdpa"
    with torch.no_grad():
        body = (transformers.LlamaModel(cfg) if spec.is_llama else transformers.GPT2Model(cfg)).eval()
    if spec.is_llama:
        body.embed_tokens = None
        layers = body.layers
    else:
        body.wte = None
        layers = body.h
    app = Flask(name)

    @app.route("/health")
    def health():
        return jsonify({"status": "healthy", "role": name, "layers": f"{start}-{end}", "model": model})

    @app.route("/process", methods=["POST"])
    def process():
        hs = torch.tensor(request.json["hidden_states"], dtype=torch.float32)
        B, S, _ = hs.shape
        with torch.no_grad():
            pos = torch.arange(S, dtype=torch.long).unsqueeze(0).expand(B, -1)
            if spec.is_llama:
                pe = body.rotary_emb(hs, pos)
                for layer in layers:
                    out = layer(hs, attention_mask=None, position_ids=pos, position_embeddings=pe)
                    hs = out[0] if isinstance(out, tuple) else out
            else:
                for layer in layers:
                    out = layer(hs)
                    hs = out[0] if isinstance(out, tuple) else out
        return jsonify({"hidden_states": hs.tolist(), "status": "success", "worker": name})

    import logging
    logging.getLogger("werkzeug").setLevel(logging.ERROR)
    app.run(host="127.0.0.1", port=port, debug=False)


def orchestrate(model: str, urls, tokens: int, temperature=0.7, top_k=50, top_p=0.9):
    import requests
    import torch.nn.functional as F
    from distributed_llm_inference_amd.models.spec import ModelSpec
    from distributed_llm_inference_amd.runtime.tokenizer import ByteTokenizer
    spec = ModelSpec.from_pretrained(model)
    torch.manual_seed(1234)
    H, V = spec.hidden_size, spec.vocab_size
    embed = torch.nn.Embedding(V, H)
    wpe = torch.nn.Embedding(spec.max_position, H) if spec.is_gpt2 else None
    norm = torch.nn.LayerNorm(H) if spec.is_gpt2 else None
    norm_w = torch.ones(H)
    lm_head = torch.nn.Linear(H, V, bias=False)
    tok = ByteTokenizer(max(V, 259))
    ids = torch.tensor([tok.encode(PROMPT)])
    headers = {"ngrok-skip-browser-warning": "true", "Content-Type": "application/json"}
    per_token, wire = [], 0
    t_all = time.time()
    with torch.no_grad():
        for step in range(tokens):
            t0 = time.time()
            hidden = embed(ids)
            if wpe is not None:
                hidden = hidden + wpe(torch.arange(ids.shape[1]))
            for u in urls:                                     # new connection per hop, as the reference
                body = json.dumps({"hidden_states": hidden.tolist()})
                wire += len(body)
                r = requests.post(f"{u}/process", data=body, headers=headers, timeout=600)
                if r.status_code != 200:
                    raise RuntimeError(r.text)
                wire += len(r.content)
                hidden = torch.tensor(r.json()["hidden_states"])
            if norm is not None:
                hidden = norm(hidden)
            else:
                hidden = hidden * torch.rsqrt(hidden.pow(2).mean(-1, keepdim=True) + spec.norm_eps) * norm_w
            logits = lm_head(hidden)                           # all positions, as the reference
            z = logits[0, -1, :] / temperature
            z[z < torch.topk(z, top_k)[0][..., -1, None]] = float("-inf")
            sl, si = torch.sort(z, descending=True)
            cp = torch.cumsum(F.softmax(sl, dim=-1), dim=-1)
            rm = cp > top_p
            rm[..., 1:] = rm[..., :-1].clone()
            rm[..., 0] = 0
            z[si[rm]] = float("-inf")
            nxt = torch.multinomial(F.softmax(z, dim=-1), 1).item()
            ids = torch.cat([ids, torch.tensor([[nxt]])], dim=1)
            per_token.append(time.time() - t0)
            print(f"  token {step + 1}/{tokens}: {per_token[-1]:.2f}s (seq {ids.shape[1]})", flush=True)
    total = time.time() - t_all
    return {"model": model, "tokens": tokens, "prompt_tokens": int(ids.shape[1] - tokens), "seconds": total,
            "tokens_per_sec": tokens / total, "p50_token_latency_s": statistics.median(per_token),
            "wire_mb_per_token": wire / tokens / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="tinyllama-1.1b")
    ap.add_argument("--tokens", type=int, default=8)
    ap.add_argument("--threads", type=int, default=0, help="torch threads per worker process (0 = default)")
    ap.add_argument("--worker", nargs=4, metavar=("START", "END", "PORT", "NAME"))
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    if a.worker:
        worker_main(a.model, int(a.worker[0]), int(a.worker[1]), int(a.worker[2]), a.worker[3])
        return
    import requests
    import socket
    from distributed_llm_inference_amd.models.partition import even_split
    from distributed_llm_inference_amd.models.spec import ModelSpec
    spec = ModelSpec.from_pretrained(a.model)
    ranges = even_split(spec.num_layers, 2)                    # 22 -> 11/11, the reference's split
    ports = []
    for _ in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            ports.append(s.getsockname()[1])
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--model", a.model, "--threads", str(a.threads),
                               "--worker", str(r[0]), str(r[1]), str(p), f"worker_{i + 1}"])
             for i, (r, p) in enumerate(zip(ranges, ports))]
    urls = [f"http://127.0.0.1:{p}" for p in ports]
    try:
        t0 = time.time()
        for u in urls:
            while True:
                try:
                    if requests.get(u + "/health", timeout=2).status_code == 200:
                        break
                except Exception:
                    if time.time() - t0 > 3600 or any(p.poll() is not None for p in procs):
                        raise RuntimeError("worker failed to start")
                    time.sleep(1.0)
        print(f"workers up after {time.time() - t0:.0f}s: layers {ranges}", flush=True)
        res = orchestrate(a.model, urls, a.tokens)
        res["hardware"] = f"host CPU, {os.cpu_count()} logical cores, orchestrator + 2 worker processes on localhost"
        res["split"] = f"{ranges[0][1] - ranges[0][0]} / {ranges[1][1] - ranges[1][0]}"
        print(json.dumps(res), flush=True)
        path = os.path.join(ROOT, "bench", "baseline_numbers.json")
        try:
            with open(path) as f:
                allr = json.load(f)
        except Exception:
            allr = {}
        allr[a.model] = res
        with open(path, "w") as f:
            json.dump(allr, f, indent=2)
    finally:
        for p in procs:
            p.terminate()
        for p in procs:
            try:
                p.wait(20)
            except Exception:
                p.kill()


if __name__ == "__main__":
    main()
