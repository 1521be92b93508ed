set -e
out=gpurun_out/r3one; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 300 --timeout-method thread -k "attention or rope" > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -3 $out/tests.log
