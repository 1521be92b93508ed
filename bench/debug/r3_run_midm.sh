# The two large-M kernels per Llama-3-8B shape over the row counts of decode micro-batches.
set -e
out=gpurun_out/r3mid; mkdir -p $out
M="${MLIST:-1536 2048 4096}"
timeout -k 10 200 python -u bench/tile_gemm_bench.py --m $M --iters 30 > $out/pingpong.txt 2>&1 || { tail $out/pingpong.txt; exit 1; }
DLIA_GEMM_LARGE=stream timeout -k 10 200 python -u bench/tile_gemm_bench.py --m $M --iters 30 > $out/stream.txt 2>&1 || { tail $out/stream.txt; exit 1; }
paste -d'|' $out/pingpong.txt $out/stream.txt | grep "M=" | sed -e 's/TFLOP\/s//g' | awk -F'|' '{print $1 "   ||stream:" substr($2, 36)}'
