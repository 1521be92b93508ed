#!/usr/bin/env bash
# Kernel-trace + (optionally) counter passes of the single-GPU decode micro-benchmark.
#
#   bench/rocprof_capture.sh OUTDIR [quick_decode.py args...]
#   PMC=1 bench/rocprof_capture.sh OUTDIR --batch 128 --layers 8
#
# Writes OUTDIR/trace (kernel trace + stats, CSV), OUTDIR/summary.txt (per-kernel medians of the
# steady state) and, with PMC=1, one directory per counter pass. Every counter pass is a run of
# its own and stays within the per-block limits of one rocprofv3 pass (8 SQ, 4 TCC, ...); the
# profiled program is started directly after `--` (no shell or env wrapper in between).
set -euo pipefail
out=${1:?usage: rocprof_capture.sh OUTDIR [quick_decode args]}; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out"; out=$(cd "$out" && pwd)
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- \
    python3 "$root/bench/quick_decode.py" --steps 20 "$@" > "$out/trace.log" 2>&1
python3 "$root/bench/prof_summary.py" "$out/trace" 1200 > "$out/summary.txt"
if [ "${PMC:-0}" = "1" ]; then
  i=0
  for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
              "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
              "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$out/pmc$i" -- \
        python3 "$root/bench/quick_decode.py" --steps 3 --warmup 1 "$@" > "$out/pmc$i.log" 2>&1
  done
fi
cat "$out/summary.txt"
