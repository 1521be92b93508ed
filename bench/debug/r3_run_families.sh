#!/bin/bash
# Llama-family dialects on the GPU: the stage tests, then bench.py on the Qwen2 and Mistral shapes.
set -u
mkdir -p gpurun_out/families
if [ -z "${SKIP_TESTS:-}" ]; then
timeout -k 10 500 python -u -m pytest tests/test_stage_gpu.py -x -q --timeout 120 --timeout-method thread -k "not real_shape" > gpurun_out/families/pytest.log 2>&1
rc=$?; tail -4 gpurun_out/families/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" gpurun_out/families/pytest.log | head -20; exit $rc; }
fi
for m in ${MODELS:-qwen2.5-3b mistral-7b}; do
  timeout -k 10 280 python bench.py --model $m --steps 16 --warmup 3 > gpurun_out/families/bench_$m.json 2> gpurun_out/families/bench_$m.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "$m failed rc=$rc"; tail -8 gpurun_out/families/bench_$m.err; exit $rc; fi
  python - gpurun_out/families/bench_$m.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["model"], d["value"], "tok/s", d["ms_per_step"], "ms/step; 64 rows", d["latency_point"]["ms_per_token"], "ms; single", d["single_request"]["ms_per_token"], "ms", flush=True)
P
done
