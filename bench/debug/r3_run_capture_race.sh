#!/bin/bash
# The 1..5-stage token-identity test on the GPU, a few times in separate processes, with the HIP runtime's error log on.
set -u
mkdir -p gpurun_out/caprace
for i in 1 2 3; do
  AMD_LOG_LEVEL=1 timeout -k 10 200 python -u -m pytest tests/test_engine_gpu.py -x -q --timeout 120 --timeout-method thread -k "pp1_to_pp5" > gpurun_out/caprace/run_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc: $(grep -c ':1:' gpurun_out/caprace/run_$i.log) runtime error lines; $(tail -1 gpurun_out/caprace/run_$i.log)"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
done
