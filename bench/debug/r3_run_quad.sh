#!/bin/bash
# The four-wave 256 x 256 GEMM against the ping-pong kernel: numerics tests, then the layer shapes at 2048 / 4096 rows.
set -u
mkdir -p gpurun_out/quad
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "gemm_large_m or gemm_pingpong_shapes" > gpurun_out/quad/tests.log 2>&1
rc=$?; tail -3 gpurun_out/quad/tests.log
[ $rc -ne 0 ] && { grep -n "assert\|Error\|FAILED" gpurun_out/quad/tests.log | head; exit $rc; }
for k in pingpong quad pingpong quad; do
  echo "== $k"
  DLIA_GEMM_LARGE=$k timeout -k 10 120 python bench/tile_gemm_bench.py --m 2048 4096 --iters 20 --no-split 2>&1 | tee -a gpurun_out/quad/bench_$k.log || exit 1
done
