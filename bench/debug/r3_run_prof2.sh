set -e
mkdir -p gpurun_out/r3p2
bash bench/rocprof_capture.sh gpurun_out/r3p2/b1 --batch 1 > gpurun_out/r3p2/c1.log 2>&1 || { tail -20 gpurun_out/r3p2/c1.log; exit 1; }
cat gpurun_out/r3p2/b1/summary.txt
bash bench/rocprof_capture.sh gpurun_out/r3p2/b64 --batch 64 > gpurun_out/r3p2/c2.log 2>&1 || { tail -20 gpurun_out/r3p2/c2.log; exit 1; }
cat gpurun_out/r3p2/b64/summary.txt
bash bench/rocprof_capture.sh gpurun_out/r3p2/tiny_b1 --batch 1 --model tinyllama-1.1b > gpurun_out/r3p2/c3.log 2>&1 || { tail -20 gpurun_out/r3p2/c3.log; exit 1; }
cat gpurun_out/r3p2/tiny_b1/summary.txt
