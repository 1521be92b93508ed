# End-of-round serving numbers (Llama-3-8B, stage process + native loop): closed loops and the HTTP front door.
set -e
out=gpurun_out/r3sf; mkdir -p $out
SB="python -u bench/serve_bench.py --isolate"
timeout -k 10 200 $SB --batch 256 --closed-loop 512 --stagger-ms 1000 --requests 4096 --prompt-min 100 --prompt-max 140 --new-tokens 32 > $out/serve_closed512.json 2> $out/serve_closed512.err
timeout -k 10 200 $SB --batch 128 --closed-loop 192 --stagger-ms 2000 --requests 3000 --prompt-min 16 --prompt-max 1200 --new-tokens 96 --max-seq-len 1536 > $out/serve_closed192_long.json 2> $out/serve_closed192_long.err
timeout -k 10 300 $SB --batch 2048 --requests 8192 --prompt-min 100 --prompt-max 140 --new-tokens 128 > $out/serve_b2048.json 2> $out/serve_b2048.err
timeout -k 10 300 python -u bench/http_bench.py --spawn --model llama-3-8b --batch 256 --micro-batches 2 --clients 512 --requests 4096 --max-tokens 32 > $out/http_512.json 2> $out/http_512.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3sf/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    keys = ["output_tok_per_s", "output_tokens_per_s", "requests_per_s", "ttft_ms_p50", "ttft_ms_p99", "tpot_ms_p50", "itl_ms_p50", "itl_ms_p99", "latency_ms_p50", "latency_ms_p99"]
    print(f.split("/")[-1], {k: (round(d[k], 2) if isinstance(d[k], float) else d[k]) for k in keys if k in d})
PY
