# TinyLlama, one request at a time, stages as threads of ONE process on one GPU (in-process device hops, native stage loop
# behind stage 0): what the pipeline machinery itself adds per token, without processes competing for the card.
set -e
out=gpurun_out/r3ip; mkdir -p $out
common="--model tinyllama-1.1b --requests 6 --prompt-min 24 --prompt-max 40 --new-tokens 96 --batch 8 --closed-loop 1"
for pp in 1 2 4; do
  timeout -k 10 250 python -u bench/serve_bench.py $common --pp-size $pp --micro-batches 2 --transport loopback > $out/tiny_inproc_pp$pp.json 2> $out/tiny_inproc_pp$pp.err
done
DLIA_KEEP_HW_QUEUES=1 timeout -k 10 250 python -u bench/serve_bench.py $common --pp-size 2 --micro-batches 2 --transport loopback > $out/tiny_inproc_pp2_q4.json 2> $out/tiny_inproc_pp2_q4.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3ip/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "tpot p50", round(d["tpot_ms_p50"], 3), "itl p50", round(d["itl_ms_p50"], 3), "ttft p50", round(d["ttft_ms_p50"], 2))
PY
