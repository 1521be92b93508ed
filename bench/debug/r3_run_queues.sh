# Hardware queues per process against the multi-stage pipeline (ranks sharing this GPU over RCCL's socket transport).
set -e
out=gpurun_out/r3q; mkdir -p $out
export DLIA_SHARE_GPU=1 DLIA_RCCL_DISTINCT_HOSTS=1
run() {
  name=$1; shift
  env $ENVV timeout -k 10 400 python -u bench.py "$@" > $out/$name.json 2> $out/$name.err || { tail -20 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
lp, sr = d.get("latency_point") or {}, d.get("single_request") or {}
print(sys.argv[2], "tok/s", d["value"], "ms/step", d["ms_per_step"], "b64", lp.get("ms_per_token"), "single", sr.get("ms_per_token"), "ranks", [r["ms_per_step"] for r in d["ranks"]], flush=True)
PY
}
for q in 4 8 16; do
ENVV="GPU_MAX_HW_QUEUES=$q"; run tiny_pp2_q$q --gpus 2 --model tinyllama-1.1b --batch 256 --steps 32 --warmup 8
done
for q in 4 8 16; do
ENVV="GPU_MAX_HW_QUEUES=$q"; run l8b_pp4_q$q --gpus 4 --batch 256 --steps 16 --warmup 4
done
