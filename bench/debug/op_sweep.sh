#!/bin/bash
# usage: op_sweep.sh "<env assignments>" <bench.py args...>   -> one line: env | mbs x size | tok/s | ms/step
envs="$1"; shift
out=$(env $envs timeout -k 10 300 python bench.py "$@") || exit 1
echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$envs', '|', c['micro_batches'], 'x', c['micro_batch_size'], '|', d['value'], 'tok/s |', d['ms_per_step'], 'ms')"
