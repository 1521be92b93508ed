mkdir -p gpurun_out/r3g
for i in 1 2; do
  for g in 0 1; do
    DLIA_GUIDED_SAMPLER=$g timeout -k 10 200 python bench.py --steps 24 --warmup 4 --latency-rows 0 > gpurun_out/r3g/bench_guided${g}_$i.json 2>/dev/null || exit 1
    python -c "import json;r=json.load(open('gpurun_out/r3g/bench_guided${g}_$i.json'));print('guided=$g run $i', r['value'], r['ms_per_step'])"
  done
done
