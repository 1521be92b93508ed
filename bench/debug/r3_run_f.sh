mkdir -p gpurun_out/r3f
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_sampler_full_gpu.py tests/test_stage_gpu.py -x -q --timeout 200 --timeout-method thread -k "sampl or head or pingpong or stack or stage" > gpurun_out/r3f/tests.log 2>&1; rc=$?; tail -15 gpurun_out/r3f/tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 4 > gpurun_out/r3f/bench_default.json 2> gpurun_out/r3f/bench_default.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r3f/bench_default.json
