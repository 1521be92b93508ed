mkdir -p gpurun_out/r3b
timeout -k 10 500 python -u -m pytest tests/test_stage_gpu.py tests/test_engine_gpu.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3b/tests.log 2>&1; rc=$?; tail -8 gpurun_out/r3b/tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
SB="python bench/serve_bench.py --batch 2048 --requests 8192 --prompt-min 100 --prompt-max 140 --new-tokens 128 --isolate"
DLIA_EAGER_BARRIER=1 timeout -k 10 200 $SB > gpurun_out/r3b/serve_b2048_barrier.json 2> gpurun_out/r3b/serve_b2048_barrier.err && tail -1 gpurun_out/r3b/serve_b2048_barrier.json | cut -c1-900 &&
timeout -k 10 200 $SB > gpurun_out/r3b/serve_b2048_permb.json 2> gpurun_out/r3b/serve_b2048_permb.err && tail -1 gpurun_out/r3b/serve_b2048_permb.json | cut -c1-900
