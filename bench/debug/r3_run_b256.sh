set -e
out=gpurun_out/r3b256; mkdir -p $out
run() {
  name=$1; shift
  env $ENVV timeout -k 10 500 python -u bench.py "$@" > $out/$name.json 2> $out/$name.err || { tail -20 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "tok/s", d["value"], "ms/step", d["ms_per_step"], flush=True)
PY
}
for i in 1 2; do
ENVV="DLIA_GPU_SPANS=0"; run A_off_$i --latency-rows 0
ENVV="DLIA_X=0"; run B_default_$i --latency-rows 0
ENVV="DLIA_SPANS_DBG=C"; run C_end_only_$i --latency-rows 0
ENVV="DLIA_SPANS_DBG=D"; run D_no_query_$i --latency-rows 0
done
