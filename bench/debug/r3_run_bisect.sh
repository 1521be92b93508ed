# Position-balanced comparison of several trees (worktrees built in place) at the default operating point.
set -e
out=$PWD/gpurun_out/r3bis; mkdir -p $out
run() {
  name=$1; dir=$2; shift; shift
  (cd $dir && timeout -k 10 500 python -u bench.py "$@" > $out/$name.json 2> $out/$name.err) || { tail -20 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "tok/s", d["value"], "ms/step", d["ms_per_step"], flush=True)
PY
}
L="--latency-rows 0"
run old_a _old; run new_a . $L; run cde9_a _w_de99e52
run cde9_b _w_de99e52; run new_b . $L; run old_b _old
run new_c . $L; run old_c _old; run new_d . $L; run cde9_c _w_de99e52
run old_d _old; run cde9_d _w_de99e52
python - <<'PY'
import json, glob
v = {}
for f in glob.glob("gpurun_out/r3bis/*_[a-d].json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    v.setdefault(f.split("/")[-1].split("_")[0], []).append(d["ms_per_step"])
for k in sorted(v): print(k, "mean ms/step", round(sum(v[k]) / len(v[k]), 3), sorted(v[k]))
PY
