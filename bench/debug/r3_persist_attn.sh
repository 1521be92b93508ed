#!/usr/bin/env bash
# Same box, alternating: default, four-wave GEMM + attention in N persistent 4-wave workgroups (one per CU beside a
# GEMM workgroup), and the controls (each change alone).
set -uo pipefail
out=gpurun_out/r3persist; mkdir -p $out
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 24 --warmup 4 --latency-rows 0 > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -5 $out/$name.err; exit 1; }
  python - <<P
import json; d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["ms_per_step"])
P
}
run default_1 X=1
run quad_persist256_1 DLIA_GEMM_LARGE=quad DLIA_ATTN_PERSIST=256
run default_2 X=1
run quad_persist256_2 DLIA_GEMM_LARGE=quad DLIA_ATTN_PERSIST=256
run pp_persist256 DLIA_ATTN_PERSIST=256
run quad_persist512 DLIA_GEMM_LARGE=quad DLIA_ATTN_PERSIST=512
run quad_persist192 DLIA_GEMM_LARGE=quad DLIA_ATTN_PERSIST=192
run pp_persist512 DLIA_ATTN_PERSIST=512
