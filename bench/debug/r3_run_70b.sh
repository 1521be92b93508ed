# per-stage cost of the Llama-3-70B 8-stage partition, one stage at a time on one GPU, and the 8-rank rehearsal
mkdir -p gpurun_out/r3_70b
timeout -k 10 500 python bench/stage_cost.py --model llama-3-70b --plan 8 --dtype fp8 --batch 64 --micro-batches 1 --streams 1 --steps 30 > gpurun_out/r3_70b/stage_cost_70b_fp8_b64_plan8.txt 2>&1 || exit 1
tail -3 gpurun_out/r3_70b/stage_cost_70b_fp8_b64_plan8.txt
timeout -k 10 500 python bench/stage_cost.py --model llama-3-70b --plan 8 --dtype bf16 --batch 2048 --steps 10 > gpurun_out/r3_70b/stage_cost_70b_bf16_2x2048_plan8.txt 2>&1 || exit 1
tail -3 gpurun_out/r3_70b/stage_cost_70b_bf16_2x2048_plan8.txt
DLIA_SHARE_GPU=1 DLIA_RCCL_DISTINCT_HOSTS=1 timeout -k 10 500 python bench.py --gpus 8 --model llama-3-70b --dtype fp8 --batch 64 --micro-batches 10 --steps 8 --warmup 2 --latency-steps 8 > gpurun_out/r3_70b/rehearsal_70b_fp8_8ranks_one_gpu.json 2> gpurun_out/r3_70b/rehearsal_70b_fp8_8ranks_one_gpu.err || { tail -20 gpurun_out/r3_70b/rehearsal_70b_fp8_8ranks_one_gpu.err; exit 1; }
cut -c1-600 gpurun_out/r3_70b/rehearsal_70b_fp8_8ranks_one_gpu.json
