mkdir -p gpurun_out/r3e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r3e/tests.log 2>&1; rc=$?; tail -8 gpurun_out/r3e/tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 4 > gpurun_out/r3e/bench_default.json 2> gpurun_out/r3e/bench_default.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r3e/bench_default.json
