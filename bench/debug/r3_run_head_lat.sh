# One request at a time (closed loop of 1 client) on TinyLlama over 2 and 3 stage processes sharing this GPU
# (RCCL's socket transport), 8 hardware queues per process (2 x 8 and 3 x 8 fit the card's queue slots; 4 x 8 do not):
# stage 0 on the Python head against the native stage-0 loop; single stage for scale.
set -e
out=gpurun_out/r3h; mkdir -p $out
export DLIA_SHARE_GPU=1 DLIA_RCCL_DISTINCT_HOSTS=1 GPU_MAX_HW_QUEUES=8
common="--model tinyllama-1.1b --requests 6 --prompt-min 24 --prompt-max 40 --new-tokens 96 --batch 8 --closed-loop 1 --isolate"
timeout -k 10 200 python -u bench/serve_bench.py $common --pp-size 1 --micro-batches 1 > $out/tiny_pp1.json 2> $out/tiny_pp1.err
for pp in 2 3; do
 for i in 1 2; do
  DLIA_NATIVE_HEAD_LOOP=0 timeout -k 10 250 python -u bench/serve_bench.py $common --pp-size $pp --micro-batches 2 --transport rccl > $out/tiny_pp${pp}_python_head_$i.json 2> $out/tiny_pp${pp}_python_head_$i.err
  DLIA_NATIVE_HEAD_LOOP=1 timeout -k 10 250 python -u bench/serve_bench.py $common --pp-size $pp --micro-batches 2 --transport rccl > $out/tiny_pp${pp}_native_head_$i.json 2> $out/tiny_pp${pp}_native_head_$i.err
 done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3h/tiny_pp*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "tpot p50", round(d["tpot_ms_p50"], 3), "itl p50", round(d["itl_ms_p50"], 3), "ttft p50", round(d["ttft_ms_p50"], 2))
PY
