# bench.py lines for the usual configurations on one GPU: bf16 / fp8 / e4m3 KV, other batch shapes, the other models.
set -e
out=gpurun_out/r3m; mkdir -p $out
run() {
  name=$1; shift
  timeout -k 10 500 python -u bench.py "$@" > $out/$name.json 2> $out/$name.err || { tail -20 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
lp, sr = d.get("latency_point") or {}, d.get("single_request") or {}
print(sys.argv[2], "tok/s", d["value"], "ms/step", d["ms_per_step"], "b64 ms", lp.get("ms_per_token"), "single ms", sr.get("ms_per_token"), flush=True)
PY
}
run bf16_default
run bf16_b1024 --batch 1024
run fp8 --dtype fp8
run fp8_kv8 --dtype fp8 --kv-dtype fp8
run bf16_kv8 --kv-dtype fp8
run gpt2 --model gpt2
run tinyllama --model tinyllama-1.1b
run b256x3 --batch 256 --micro-batches 3 --decode-streams 3
run l70b_bf16 --model llama-3-70b --batch 1024 --steps 16 --warmup 2
run l70b_fp8_b64 --model llama-3-70b --dtype fp8 --batch 64 --micro-batches 1 --steps 32 --warmup 4
