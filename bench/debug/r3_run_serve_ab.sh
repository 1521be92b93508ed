set -e
out=gpurun_out/r3sab; mkdir -p $out
SB="python -u bench/serve_bench.py --isolate"
run() {
  name=$1; shift
  timeout -k 10 300 $SB "$@" > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["output_tok_per_s"], d["requests_per_s"], "ttft p50", round(d["ttft_ms_p50"], 1), "tpot p50", round(d["tpot_ms_p50"], 1), "itl p50/p99", round(d["itl_ms_p50"], 1), round(d["itl_ms_p99"], 1), d["steps"], flush=True)
PY
}
B2048="--batch 2048 --requests 8192 --prompt-min 100 --prompt-max 140 --new-tokens 128"
C512="--batch 256 --closed-loop 512 --stagger-ms 1000 --requests 4096 --prompt-min 100 --prompt-max 140 --new-tokens 32"
C192="--batch 128 --closed-loop 192 --stagger-ms 2000 --requests 3000 --prompt-min 16 --prompt-max 1200 --new-tokens 96 --max-seq-len 1536"
run b2048_mixed1 $B2048 --mixed-prefill 1
run b2048_mixed0 $B2048 --mixed-prefill 0
run c512_mixed1 $C512 --mixed-prefill 1
run c512_mixed0 $C512 --mixed-prefill 0
run c192_mixed1 $C192 --mixed-prefill 1
run c192_mixed0 $C192 --mixed-prefill 0
