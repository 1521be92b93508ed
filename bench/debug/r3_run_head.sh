set -e
mkdir -p gpurun_out/r3h
timeout -k 10 900 python -u -m pytest tests/test_engine_gpu.py tests/test_bench_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3h/tests.log 2>&1 || { tail -60 gpurun_out/r3h/tests.log; exit 1; }
tail -5 gpurun_out/r3h/tests.log
