set -e
mkdir -p gpurun_out/s2
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_stage_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3
run() { # label, lib, cmd...
  local label=$1 lib=$2; shift 2
  if [ "$lib" = old ]; then export DLIA_OPS_LIB=$PWD/distributed_llm_inference_amd/ops/_C_old.so; else unset DLIA_OPS_LIB; fi
  echo -n "$label $lib: "; timeout -k 10 200 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value', d.get('tok_per_s')), d.get('ms_per_step'))"
}
for v in old new old new; do run "bf16 3x256" $v python bench.py --steps 30 --warmup 5 --batch 256 --micro-batches 3 --decode-streams 3; done
for v in old new; do run "fp8 3x256" $v python bench.py --steps 30 --warmup 5 --batch 256 --micro-batches 3 --decode-streams 3 --dtype fp8; done
for v in old new old new; do run "fp8 quick b64" $v python bench/quick_decode.py --dtype fp8 --batch 64; done
for v in old new; do run "fp8 quick b128" $v python bench/quick_decode.py --dtype fp8 --batch 128; done
for v in old new; do run "bf16 quick b256" $v python bench/quick_decode.py --batch 256; done
for v in old new; do run "fp8w 3x256" $v python bench.py --steps 30 --warmup 5 --dtype fp8w; done
