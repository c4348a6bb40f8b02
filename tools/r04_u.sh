# smoke() and the default bench line with the final binary and the final traffic file
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench.log 2>&1 || { tail -20 $O/bench.log; exit 1; }
tail -1 $O/bench.log | cut -c1-300
