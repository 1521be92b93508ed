# Launch-latency switches of the HIP runtime against the low-latency regions of bench.py (same box, alternating).
set -e
out=gpurun_out/r3env; mkdir -p $out
run() {  # name, env assignments...
  name=$1; shift
  env "$@" timeout -k 10 200 python -u bench.py --model tinyllama-1.1b --steps 8 --warmup 2 --latency-steps 200 > $out/$name.json 2> $out/$name.err || { tail -20 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "tok/s", d["value"], "b64 ms", d["latency_point"]["ms_per_token"], "single ms", d["single_request"]["ms_per_token"], flush=True)
PY
}
run base_1 DLIA_X=0
run devkernarg_1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0_1 HIP_FORCE_DEV_KERNARG=0
run pktcap_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run pktcap0_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run base_2 DLIA_X=0
run devkernarg_2 HIP_FORCE_DEV_KERNARG=1
