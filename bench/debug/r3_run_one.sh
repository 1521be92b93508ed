set -e
out=gpurun_out/r3one; mkdir -p $out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -3 $out/tests.log
bash bench/rocprof_capture.sh gpurun_out/r3one/b256 --batch 256 --layers 8 > $out/c1.log 2>&1 || { tail -20 $out/c1.log; exit 1; }
grep -E "sample_kernel|774144|kernels " gpurun_out/r3one/b256/summary.txt
