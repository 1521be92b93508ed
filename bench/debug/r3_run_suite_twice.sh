#!/bin/bash
# The GPU suite as the driver runs it, several times in separate processes (looking for tests that fail now and then).
set -u
mkdir -p gpurun_out/suite
for i in 1 2 3; do
  timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/suite/run_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc: $(tail -1 gpurun_out/suite/run_$i.log)"
  if [ $rc -ne 0 ]; then grep -n "^FAILED\|^ERROR\|Error" gpurun_out/suite/run_$i.log | head -10; exit $rc; fi
done
