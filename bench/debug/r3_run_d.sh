mkdir -p gpurun_out/r3d
timeout -k 10 600 python -u -m pytest tests/test_engine_gpu.py tests/test_bench_gpu.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3d/tests.log 2>&1; rc=$?; tail -30 gpurun_out/r3d/tests.log; echo "tests rc=$rc"
