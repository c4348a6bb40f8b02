# update_weights through the reciprocal (enoki array / scalar): MIS parity subset + C5SM / C3M bench lines
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "mis or spectral or bins" > $O/p_tests.log 2>&1 || { tail -30 $O/p_tests.log; exit 1; }
tail -2 $O/p_tests.log
for c in C5SM C3M; do timeout -k 10 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/p_bench_$c.log 2>&1; tail -1 $O/p_bench_$c.log | cut -c1-160; done
