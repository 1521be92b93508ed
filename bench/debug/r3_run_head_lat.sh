# One request at a time (closed loop of 1 client) on TinyLlama over 2 and 4 stage processes sharing this GPU
# (RCCL's socket transport): stage 0 on the Python head against the native stage-0 loop; single stage for scale.
set -e
out=gpurun_out/r3h; mkdir -p $out
export DLIA_SHARE_GPU=1 DLIA_RCCL_DISTINCT_HOSTS=1
common="--model tinyllama-1.1b --requests 6 --prompt-min 24 --prompt-max 40 --new-tokens 96 --batch 8 --closed-loop 1 --isolate"
timeout -k 10 200 python -u bench/serve_bench.py $common --pp-size 1 --micro-batches 1 > $out/tiny_pp1.json 2> $out/tiny_pp1.err
for pp in 2 4; do
  DLIA_NATIVE_HEAD_LOOP=0 timeout -k 10 250 python -u bench/serve_bench.py $common --pp-size $pp --micro-batches 2 --transport rccl > $out/tiny_pp${pp}_python_head.json 2> $out/tiny_pp${pp}_python_head.err
  DLIA_NATIVE_HEAD_LOOP=1 timeout -k 10 250 python -u bench/serve_bench.py $common --pp-size $pp --micro-batches 2 --transport rccl > $out/tiny_pp${pp}_native_head.json 2> $out/tiny_pp${pp}_native_head.err
done
tail -n 2 $out/tiny_pp*.json
