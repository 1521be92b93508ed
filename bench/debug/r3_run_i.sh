mkdir -p gpurun_out/r3i
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py tests/test_sampler_full_gpu.py -x -q --timeout 200 --timeout-method thread -k "sampl or head" > gpurun_out/r3i/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r3i/tests.log
[ $rc -eq 0 ] || exit $rc
bash bench/rocprof_capture.sh gpurun_out/r3i/b2048 --batch 2048 --layers 8 > gpurun_out/r3i/c1.log 2>&1 || exit 1
grep -E "sample_kernel|kernels " gpurun_out/r3i/b2048/summary.txt
bash bench/rocprof_capture.sh gpurun_out/r3i/b1 --batch 1 > gpurun_out/r3i/c2.log 2>&1 || exit 1
cat gpurun_out/r3i/b1/summary.txt
bash bench/rocprof_capture.sh gpurun_out/r3i/b64 --batch 64 > gpurun_out/r3i/c3.log 2>&1 || exit 1
cat gpurun_out/r3i/b64/summary.txt
