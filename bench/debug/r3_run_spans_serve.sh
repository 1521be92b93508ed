# Device-side step timing on / off against throughput: tests first, then bench.py (Python loop) and serve_bench (native loop).
set -e
out=gpurun_out/r3ss; mkdir -p $out
timeout -k 10 900 python -u -m pytest tests/test_engine_gpu.py tests/test_local_edge_gpu.py tests/test_rccl_gpu.py tests/test_bench_gpu.py -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -3 $out/tests.log
run() {
  name=$1; shift
  env $ENVV timeout -k 10 500 python -u bench.py "$@" > $out/$name.json 2> $out/$name.err || { tail -20 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "tok/s", d["value"], "ms/step", d["ms_per_step"], flush=True)
PY
}
A="--batch 256 --micro-batches 3 --decode-streams 3 --latency-rows 0"
ENVV="DLIA_X=0"; run b256_default $A
ENVV="DLIA_GPU_SPANS=0"; run b256_off $A
ENVV="DLIA_X=0"; run big_default --latency-rows 0
ENVV="DLIA_GPU_SPANS=0"; run big_off --latency-rows 0
ENVV="DLIA_X=0"; run big_default2 --latency-rows 0
ENVV="DLIA_GPU_SPANS=0"; run big_off2 --latency-rows 0
common="--batch 256 --micro-batches 3 --requests 3072 --prompt-min 100 --prompt-max 140 --new-tokens 128 --isolate"
for m in 1 0 1 0; do
  DLIA_GPU_SPANS=$m timeout -k 10 300 python -u bench/serve_bench.py $common > $out/sat_mode_$m.json 2> $out/sat_mode_$m.err
  python - $out/sat_mode_$m.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], d["output_tok_per_s"], d["requests_per_s"], "tpot p50", round(d["tpot_ms_p50"], 2), "itl p50", round(d["itl_ms_p50"], 2))
PY
done
