"""Command line: `python -m distributed_llm_inference_amd <command>`.

    serve     orchestrator API over a pipeline of N stages (or over HTTP workers)
    worker    one reference-compatible HTTP layer worker (/process, /health)
    chat      interactive client (the reference's Test.py)
    generate  one prompt through a local in-process engine
    bench     the headline decode benchmark (same as bench.py)
    plan      print the layer partition for a model / stage count
"""
from __future__ import annotations

import argparse
import json
import os
import sys

# (GPU_MAX_HW_QUEUES and HSA_ENABLE_IPC_MODE_LEGACY are settled when the package is imported: __init__.py)

from ..config import EngineConfig


def _add_engine_args(ap):
    ap.add_argument("--config", help="YAML file with EngineConfig fields")
    ap.add_argument("--model")
    ap.add_argument("--dtype", choices=["bf16", "fp8", "fp8w", "fp32"], help="fp8 = layer GEMMs on the fp8 MFMA; fp8w = e4m3 weights only")
    ap.add_argument("--kv-dtype", choices=["bf16", "fp8"], help="fp8 = e4m3 KV cache with one scale per token and kv head (HIP backend)")
    ap.add_argument("--pp-size", type=int)
    ap.add_argument("--partition", choices=["balanced", "even", "explicit"])
    ap.add_argument("--layer-ranges", help='explicit ranges, e.g. "0-11,11-22"')
    ap.add_argument("--whole-layers", action="store_true",
                    help="balanced partition over whole layers only (default: boundaries may fall between a layer's attention and MLP blocks)")
    ap.add_argument("--backend", choices=["auto", "hip", "torch"])
    ap.add_argument("--device", choices=["auto", "cuda", "cpu"])
    ap.add_argument("--transport", choices=["auto", "rccl", "rccl-torch", "gloo", "loopback", "http"],
                    help="rccl: native ncclSend/ncclRecv hops over xGMI (default on GPUs); rccl-torch: torch.distributed p2p")
    ap.add_argument("--isolate", action="store_true",
                    help="run every stage (also a single one) in its own process: the serving process then holds no GPU context")
    ap.add_argument("--kv-mem-fraction", type=float, help="share of free device memory the KV cache takes (default 0.85)")
    ap.add_argument("--server", choices=["uvicorn", "flask"], help="HTTP server for `serve` (default uvicorn)")
    ap.add_argument("--max-batch", type=int)
    ap.add_argument("--num-micro-batches", type=int)
    ap.add_argument("--max-seq-len", type=int)
    ap.add_argument("--prefill-chunk", type=int)
    ap.add_argument("--kv-blocks", type=int)
    ap.add_argument("--max-tokens-cap", type=int)
    ap.add_argument("--decode-buckets", help='row counts below --max-batch with a captured decode step of their own, e.g. "8,64": '
                                             "with few sequences running a step covers that many rows instead of all")
    ap.add_argument("--max-queue", type=int, help="requests allowed to wait for a row before new ones are refused (0 = unbounded)")
    ap.add_argument("--request-timeout-s", type=float, help="server-side deadline of a request; it is cancelled when it passes")
    ap.add_argument("--host")
    ap.add_argument("--port", type=int)
    ap.add_argument("--worker-urls", help="comma-separated worker URLs (HTTP relay mode)")
    ap.add_argument("--seed", type=int)
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--debug-top5", action="store_true")
    ap.add_argument("--auto-restart", action="store_true", help="rebuild the pipeline after a fatal stage failure")
    ap.add_argument("--fault-inject")
    ap.add_argument("--trace", dest="trace_path", help="write a Chrome-trace JSON of the step timelines here")


def config_from_args(a) -> EngineConfig:
    base = EngineConfig.from_yaml(a.config).to_dict() if getattr(a, "config", None) else {}
    base = EngineConfig.from_env(EngineConfig.from_dict(base)).to_dict()
    for k in ("model", "dtype", "kv_dtype", "pp_size", "partition", "backend", "device", "transport", "max_batch", "num_micro_batches",
              "max_seq_len", "prefill_chunk", "kv_blocks", "max_tokens_cap", "host", "port", "seed", "fault_inject",
              "trace_path", "kv_mem_fraction", "server", "max_queue", "request_timeout_s"):
        v = getattr(a, k, None)
        if v is not None:
            base[k] = v
    if getattr(a, "layer_ranges", None):
        base["layer_ranges"] = [tuple(int(x) for x in p.split("-")) for p in a.layer_ranges.split(",")]
    if getattr(a, "decode_buckets", None):
        base["decode_buckets"] = [int(x) for x in a.decode_buckets.split(",") if x.strip()]
    if getattr(a, "worker_urls", None):
        base["worker_urls"] = [u for u in a.worker_urls.split(",") if u]
    if getattr(a, "no_graphs", False):
        base["use_graphs"] = False
    if getattr(a, "debug_top5", False):
        base["debug_top5"] = True
    if getattr(a, "auto_restart", False):
        base["auto_restart"] = True
    if getattr(a, "isolate", False):
        base["isolate"] = True
    if getattr(a, "whole_layers", False):
        base["split_layers"] = False
    if getattr(a, "pp_size", None) is not None and getattr(a, "num_micro_batches", None) is None:
        base["num_micro_batches"] = 0
    return EngineConfig.from_dict(base)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="distributed_llm_inference_amd", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("serve"); _add_engine_args(p)
    p = sub.add_parser("worker"); _add_engine_args(p)
    p.add_argument("--index", type=int, required=True, help="0-based worker index")
    p.add_argument("--num-workers", type=int, default=2)
    p = sub.add_parser("chat"); p.add_argument("url", nargs="?")
    p.add_argument("--stream", action="store_true", help="print each answer while it is decoded (server-sent events)")
    p = sub.add_parser("generate"); _add_engine_args(p)
    p.add_argument("prompt"); p.add_argument("--max-tokens", type=int, default=20)
    p.add_argument("--temperature", type=float, default=None)
    p = sub.add_parser("plan"); _add_engine_args(p)
    p = sub.add_parser("bench"); p.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)

    if a.cmd == "serve":
        from .orchestrator import serve
        serve(config_from_args(a))
    elif a.cmd == "worker":
        from .worker import serve_worker
        cfg = config_from_args(a)
        serve_worker(cfg, a.index, a.num_workers, port=a.port)
    elif a.cmd == "chat":
        from .client import main as client_main
        return client_main(([a.url] if a.url else []) + (["--stream"] if a.stream else []))
    elif a.cmd == "generate":
        from ..parallel.launcher import make_pipeline
        cfg = config_from_args(a)
        pipe = make_pipeline(cfg)
        try:
            print(json.dumps(pipe.engine.generate_text(a.prompt, a.max_tokens, a.temperature), indent=2))
        finally:
            pipe.close()
    elif a.cmd == "plan":
        from ..parallel.launcher import plans_for, resolve_spec
        cfg = config_from_args(a)
        spec = resolve_spec(cfg)
        for pl in plans_for(cfg, spec):
            extra = [x for x, on in (("embedding", pl.has_embed), ("final norm + lm_head + sampler", pl.has_head)) if on]
            # x.5 = a boundary between a layer's attention and MLP blocks (`StagePlan.describe`)
            print(f"stage {pl.stage}: layers {pl.describe()}" + (f" + {', '.join(extra)}" if extra else ""))
    elif a.cmd == "bench":
        import os
        import runpy
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        sys.argv = ["bench.py"] + a.rest
        runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
    return 0
