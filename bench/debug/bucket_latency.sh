#!/usr/bin/env bash
# one client through the HTTP API on servers sized for throughput, with and without decode buckets
set -uo pipefail
out=gpurun_out/bucket; mkdir -p $out
for cfg in "256 -" "256 8,64" "2048 -" "2048 8,64,512"; do
  set -- $cfg
  extra=""; [ "$2" != "-" ] && extra="--decode-buckets $2"
  timeout -k 10 300 python bench/http_bench.py --spawn --model llama-3-8b --batch $1 --micro-batches 2 --clients 1 --requests 6 --max-tokens 64 --max-seq-len 512 --show-one $extra > $out/lat_$1_${2//,/_}.log 2>&1 || { echo "failed $cfg"; tail -5 $out/lat_$1_${2//,/_}.log; exit 1; }
  echo "batch=$1 buckets=$2 $(grep one_generate $out/lat_$1_${2//,/_}.log | python -c "import sys,json; d=json.loads(sys.stdin.read())['one_generate']; print('ttft_ms', d['ttft_ms'], 'tpot_ms', d['tpot_ms_p50'], 'time', d['time_taken'])")"
done
# and a loaded server is no slower with buckets configured
timeout -k 10 300 python bench/http_bench.py --spawn --model llama-3-8b --batch 256 --micro-batches 2 --clients 512 --requests 2048 --max-tokens 64 --max-seq-len 512 --decode-buckets 8,64 > $out/load_256_buckets.log 2>&1 || { echo "load failed"; tail -5 $out/load_256_buckets.log; exit 1; }
tail -1 $out/load_256_buckets.log | cut -c1-200
