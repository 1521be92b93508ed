mkdir -p gpurun_out/r3_prof gpurun_out/r3_http
PMC=1 bash bench/rocprof_capture.sh gpurun_out/r3_prof/b2048 --batch 2048 --layers 8 > gpurun_out/r3_prof/capture.log 2>&1 || { tail -20 gpurun_out/r3_prof/capture.log; exit 1; }
head -14 gpurun_out/r3_prof/b2048/summary.txt
timeout -k 10 300 python bench/http_bench.py --spawn --batch 256 --micro-batches 2 --clients 512 --requests 4096 --max-tokens 32 > gpurun_out/r3_http/http_8b_512_clients.json 2> gpurun_out/r3_http/http_8b_512_clients.err || { tail gpurun_out/r3_http/http_8b_512_clients.err; exit 1; }
tail -1 gpurun_out/r3_http/http_8b_512_clients.json | cut -c1-700
timeout -k 10 300 python bench/http_bench.py --spawn --batch 256 --micro-batches 2 --clients 1 --requests 6 --max-tokens 64 --show-one > gpurun_out/r3_http/http_8b_one_client.json 2>/dev/null; tail -2 gpurun_out/r3_http/http_8b_one_client.json | cut -c1-500
timeout -k 10 300 python bench.py --batch 1024 --steps 24 > gpurun_out/r3_prof/bench_2x1024.json 2>/dev/null; cut -c1-200 gpurun_out/r3_prof/bench_2x1024.json
timeout -k 10 300 python bench.py --model tinyllama-1.1b --steps 24 > gpurun_out/r3_prof/bench_tinyllama.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r3_prof/bench_tinyllama.json')); print(r['value'], r['ms_per_step'], r['latency_point'], r['single_request'])"
