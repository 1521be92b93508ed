#!/bin/bash
# Epilogue changes (fast reciprocal in SiLU / GELU, residuals loaded ahead): the large-M GEMMs of this tree (B) against
# a library whose gemm_pingpong.hip was compiled from the previous sources (A = $LIB_A), bf16 and e4m3, alternating.
set -u
mkdir -p gpurun_out/epi
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "gemm" > gpurun_out/epi/tests.log 2>&1
rc=$?; tail -2 gpurun_out/epi/tests.log; [ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  for v in A B; do
    lib=""; [ $v = A ] && lib="DLIA_OPS_LIB=$LIB_A"
    echo "== $v bf16"; env $lib timeout -k 10 120 python bench/tile_gemm_bench.py --m 2048 4096 --iters 20 2>&1 | grep -v amdgpu.ids | grep "layer\|gate_up\|down" || exit 1
    echo "== $v a8w8"; env $lib timeout -k 10 120 python bench/tile_gemm_bench.py --m 2048 4096 --iters 20 --a8w8 2>&1 | grep -v amdgpu.ids | grep "layer" || exit 1
  done
done
