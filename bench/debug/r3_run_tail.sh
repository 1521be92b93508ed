set -e
out=gpurun_out/r3tail; mkdir -p $out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -3 $out/tests.log
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python -u bench.py $BARGS > $out/$name.json 2> $out/$name.err || { tail -20 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "tok/s", d["value"], "ms/step", d["ms_per_step"], "b64 ms", d["latency_point"]["ms_per_token"], "single ms", d["single_request"]["ms_per_token"], flush=True)
PY
}
BARGS="--latency-steps 100"
run new_1 DLIA_X=0
run new_2 DLIA_X=0
BARGS="--model tinyllama-1.1b --steps 8 --warmup 2 --latency-steps 200"
run tiny_new_1 DLIA_X=0
run tiny_new_2 DLIA_X=0
