mkdir -p gpurun_out/r3c
timeout -k 10 300 python -u -m pytest tests/test_engine_gpu.py -x -q --timeout 200 --timeout-method thread -k "native or stage_process or rccl or dying" > gpurun_out/r3c/tests.log 2>&1; rc=$?; tail -4 gpurun_out/r3c/tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
SB="python bench/serve_bench.py --batch 2048 --requests 8192 --prompt-min 100 --prompt-max 140 --new-tokens 128 --isolate"
DLIA_EAGER_BARRIER=1 timeout -k 10 200 $SB > gpurun_out/r3c/serve_b2048_barrier.json 2> gpurun_out/r3c/serve_b2048_barrier.err && tail -1 gpurun_out/r3c/serve_b2048_barrier.json | cut -c1-420 &&
timeout -k 10 200 $SB > gpurun_out/r3c/serve_b2048_permb.json 2> gpurun_out/r3c/serve_b2048_permb.err && tail -1 gpurun_out/r3c/serve_b2048_permb.json | cut -c1-420 &&
timeout -k 10 200 $SB --prefill-chunk 2048 > gpurun_out/r3c/serve_b2048_permb_c2048.json 2> gpurun_out/r3c/serve_b2048_permb_c2048.err && tail -1 gpurun_out/r3c/serve_b2048_permb_c2048.json | cut -c1-420
