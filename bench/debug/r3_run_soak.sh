# Soaks of the HTTP front door with impatient clients after this state of the tree: one stage, and two stage
# processes over RCCL hops (sharing the GPU) with the native loops on both stages.
set -e
out=gpurun_out/r3soak; mkdir -p $out
timeout -k 10 300 python -u bench/abort_soak.py --model llama-3-8b --batch 256 --clients 256 --seconds 45 > $out/soak_pp1.json 2> $out/soak_pp1.err || { tail -20 $out/soak_pp1.err; exit 1; }
tail -n 1 $out/soak_pp1.json | cut -c1-600
DLIA_SHARE_GPU=1 DLIA_RCCL_DISTINCT_HOSTS=1 timeout -k 10 400 python -u bench/abort_soak.py --model llama-3-8b --pp-size 2 --transport rccl --micro-batches 3 --batch 128 --clients 128 --seconds 45 > $out/soak_pp2_rccl.json 2> $out/soak_pp2_rccl.err || { tail -20 $out/soak_pp2_rccl.err; exit 1; }
tail -n 1 $out/soak_pp2_rccl.json | cut -c1-600
