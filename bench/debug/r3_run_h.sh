mkdir -p gpurun_out/r3h gpurun_out/r3_70b
timeout -k 10 700 python -u -m pytest tests/test_engine_gpu.py tests/test_stage_gpu.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3h/tests.log 2>&1; rc=$?; tail -6 gpurun_out/r3h/tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
SB="python bench/serve_bench.py --isolate"
for mixed in 0 1; do
  timeout -k 10 200 $SB --mixed-prefill $mixed --batch 256 --closed-loop 512 --stagger-ms 1000 --requests 4096 --prompt-min 100 --prompt-max 140 --new-tokens 32 > gpurun_out/r3h/serve_closed512_mixed$mixed.json 2>/dev/null || exit 1
  tail -1 gpurun_out/r3h/serve_closed512_mixed$mixed.json | cut -c60-520
  timeout -k 10 200 $SB --mixed-prefill $mixed --batch 128 --closed-loop 192 --stagger-ms 2000 --requests 3000 --prompt-min 16 --prompt-max 1200 --new-tokens 96 --max-seq-len 1536 > gpurun_out/r3h/serve_closed192_long_mixed$mixed.json 2>/dev/null || exit 1
  tail -1 gpurun_out/r3h/serve_closed192_long_mixed$mixed.json | cut -c60-520
done
DLIA_SHARE_GPU=1 DLIA_RCCL_DISTINCT_HOSTS=1 timeout -k 10 500 python bench.py --gpus 8 --model llama-3-70b --dtype fp8 --batch 64 --micro-batches 10 --steps 8 --warmup 2 --latency-steps 8 > gpurun_out/r3_70b/rehearsal_70b_fp8_8ranks_one_gpu.json 2> gpurun_out/r3_70b/rehearsal_70b_fp8_8ranks_one_gpu.err || { tail -20 gpurun_out/r3_70b/rehearsal_70b_fp8_8ranks_one_gpu.err; exit 1; }
cut -c1-400 gpurun_out/r3_70b/rehearsal_70b_fp8_8ranks_one_gpu.json
