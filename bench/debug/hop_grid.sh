# RCCL point-to-point kernel shapes (workgroups = channels) and self-hop bandwidth under channel limits
out=$GRAFT_REPO_ROOT/gpurun_out/s2/hop
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for ch in default 1 2 4; do
  if [ $ch != default ]; then export NCCL_MAX_P2P_NCHANNELS=$ch NCCL_MIN_P2P_NCHANNELS=1; fi
  rm -rf $out/t_$ch
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/t_$ch -- python3 $GRAFT_REPO_ROOT/bench/hop_bench.py --iters 20 > $out/out_$ch.txt 2>&1
  echo "== channels limit: $ch"
  grep rccl_version $out/out_$ch.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['rows']: print('  rows',r['rows'],'hidden',r['hidden'],'bytes',r['bytes'],'device us/hop',r['device_us_per_hop'],'GB/s',r['GBps'])
"
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/t_$ch/**/*_kernel_trace.csv",recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "rccl" in r["Kernel_Name"]:
        agg[int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k in sorted(agg): print("  rccl kernel workgroups",k,"launches",len(agg[k]),"median us",sorted(agg[k])[len(agg[k])//2])
PY
  rm -rf $out/t_$ch
done
