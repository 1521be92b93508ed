#!/bin/bash
# bench.py at a list of operating points on one box: each argument is a string of bench.py flags ("-" = defaults).
set -u
mkdir -p gpurun_out/points
i=0
for flags in "$@"; do
  i=$((i + 1))
  [ "$flags" = "-" ] && flags=""
  out=gpurun_out/points/run_$i.json
  timeout -k 10 280 python bench.py --latency-rows 0 $flags > $out 2> gpurun_out/points/run_$i.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "run $i [$flags] failed rc=$rc"; tail -5 gpurun_out/points/run_$i.err; exit $rc; fi
  python - "$out" "$flags" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print(f"[{sys.argv[2] or 'defaults'}] {d['value']:.0f} tok/s  {d['ms_per_step']} ms/step  p50 {d['p50_token_latency_ms']} ms  {c['micro_batches']} x {c['micro_batch_size']} on {c['decode_streams']} streams", flush=True)
P
done
