set -e
out=gpurun_out/r3one; mkdir -p $out
timeout -k 10 900 python -u -m pytest tests/test_engine_gpu.py tests/test_bench_gpu.py -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -3 $out/tests.log
bash bench/debug/r3_run_inproc_pp.sh
