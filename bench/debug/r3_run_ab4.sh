# Position-balanced A/B of two environments of bench.py on one box (consecutive launches alternate fast / slow readings).
set -e
out=$PWD/gpurun_out/r3ab4; mkdir -p $out
run() {
  name=$1; shift
  env $1 timeout -k 10 500 python -u bench.py --latency-rows 0 $BARGS > $out/$name.json 2> $out/$name.err || { tail -20 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "tok/s", d["value"], "ms/step", d["ms_per_step"], flush=True)
PY
}
A="${ENV_A:-DLIA_X=0}"; B="${ENV_B:-DLIA_GPU_SPANS=0}"
run A1 $A; run B1 $B; run B2 $B; run A2 $A; run A3 $A; run B3 $B; run B4 $B; run A4 $A
python - <<'PY'
import json, glob
v = {}
for f in glob.glob("gpurun_out/r3ab4/[AB][1-4].json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    v.setdefault(f.split("/")[-1][0], []).append(d["ms_per_step"])
for k in sorted(v): print(k, "mean ms/step", round(sum(v[k]) / len(v[k]), 3), sorted(v[k]))
PY
