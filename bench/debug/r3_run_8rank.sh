# bench.py with 8 ranks sharing this GPU (RCCL socket transport), both launchers: the exact N = 8 command shapes, small batch.
set -e
out=gpurun_out/r3r8; mkdir -p $out
export DLIA_SHARE_GPU=1 DLIA_RCCL_DISTINCT_HOSTS=1
timeout -k 10 500 python -u bench.py --gpus 8 --batch 64 --steps 8 --warmup 2 --latency-steps 8 > $out/self_8.json 2> $out/self_8.err || { tail -30 $out/self_8.err; exit 1; }
python - $out/self_8.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("self-launch:", d["n_gpus"], d["value"], d["config"]["launcher"], d["config"]["transport"], [r["native_loop_steps"] for r in d["ranks"]], d["config"]["layer_ranges"])
PY
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --batch 64 --steps 8 --warmup 2 --latency-steps 8 > $out/torchrun_8.json 2> $out/torchrun_8.err || { tail -30 $out/torchrun_8.err; exit 1; }
python - $out/torchrun_8.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("torchrun:", d["n_gpus"], d["value"], d["config"]["launcher"], d["config"]["transport"], [r["native_loop_steps"] for r in d["ranks"]], [h["us_p50"] for h in d["hops"]][:4])
PY
