#!/usr/bin/env bash
# A/B on one box, alternating: the default eight-wave GEMM against the four-wave one (368 registers, one wave
# per SIMD: leaves room for an attention wave of the other stream on the same CU) in bench.py's two-stream regime.
set -uo pipefail
out=gpurun_out/r3quad; mkdir -p $out
for i in 1 2; do
  for k in default quad; do
    if [ $k = quad ]; then export DLIA_GEMM_LARGE=quad; else unset DLIA_GEMM_LARGE; fi
    timeout -k 10 200 python bench.py --steps 24 --warmup 4 --latency-rows 0 > $out/${k}_$i.json 2> $out/${k}_$i.err || { echo "$k $i failed"; tail -5 $out/${k}_$i.err; exit 1; }
    python - <<P
import json; d=json.loads(open("$out/${k}_$i.json").read().strip().splitlines()[-1]); print("$k $i", d["value"], d["ms_per_step"])
P
  done
done
