set -e
out=gpurun_out/r3full; mkdir -p $out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -60 $out/tests.log; exit 1; }
tail -3 $out/tests.log
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -30 $out/smoke.log; exit 1; }
tail -2 $out/smoke.log
timeout -k 10 300 python -u bench.py > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
cat $out/bench.json
