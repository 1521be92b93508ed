set -e
mkdir -p gpurun_out/r3p256
bash bench/rocprof_capture.sh gpurun_out/r3p256/b256 --batch 256 --layers 8 > gpurun_out/r3p256/c1.log 2>&1 || { tail -20 gpurun_out/r3p256/c1.log; exit 1; }
cat gpurun_out/r3p256/b256/summary.txt
bash bench/rocprof_capture.sh gpurun_out/r3p256/b128 --batch 128 --layers 8 > gpurun_out/r3p256/c2.log 2>&1 || { tail -20 gpurun_out/r3p256/c2.log; exit 1; }
cat gpurun_out/r3p256/b128/summary.txt
bash bench/rocprof_capture.sh gpurun_out/r3p256/b512 --batch 512 --layers 8 > gpurun_out/r3p256/c3.log 2>&1 || { tail -20 gpurun_out/r3p256/c3.log; exit 1; }
cat gpurun_out/r3p256/b512/summary.txt
