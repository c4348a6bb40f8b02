set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
for c in C1 C1L C1W; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$c.log 2>&1; tail -1 $O/bench_$c.log | cut -c1-160; done
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
