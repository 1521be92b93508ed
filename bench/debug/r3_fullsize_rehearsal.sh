#!/usr/bin/env bash
# Multi-rank rehearsal at the benchmark's real micro-batch size (2048 rows: 16.8 MB per hop, 8 KB of
# token ids back, 2048-row descriptors) with the ranks sharing the one GPU over RCCL's socket transport.
set -uo pipefail
out=gpurun_out/r3fullsize; mkdir -p $out
export DLIA_SHARE_GPU=1 DLIA_RCCL_DISTINCT_HOSTS=1
[ -n "${SKIP_N2:-}" ] || timeout -k 10 420 python bench.py --gpus 2 --batch 2048 --micro-batches 4 --steps 8 --warmup 2 --latency-rows 0 --max-seconds 400 \
    > $out/n2_self.json 2> $out/n2_self.err || { echo "n2 failed"; tail -20 $out/n2_self.err; exit 1; }
[ -n "${SKIP_N2:-}" ] || cut -c1-400 $out/n2_self.json
timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 4 --batch 2048 --micro-batches 3 --steps 8 --warmup 2 --latency-rows 0 --max-seconds 400 \
    > $out/n4_torchrun.json 2> $out/n4_torchrun.err || { echo "n4 failed"; tail -20 $out/n4_torchrun.err; exit 1; }
cut -c1-400 $out/n4_torchrun.json
