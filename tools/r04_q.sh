# spectral `path` as the flat loop + film slots per pass: full GPU suite, then C1S flat against nested, C3 and C5SB lines
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "passes or spectral or bins or bin_ or cpp or sigint or scaling or rehearsal" > $O/q_tests.log 2>&1 || { tail -30 $O/q_tests.log; exit 1; }
tail -2 $O/q_tests.log
timeout -k 10 300 python bench.py --config C1S --steps 3 --warmup 1 --no-cpu-baseline > $O/q_bench_C1S.log 2>&1; tail -1 $O/q_bench_C1S.log | cut -c1-160
MTSAMD_KERNEL=nested timeout -k 10 300 python bench.py --config C1S --steps 3 --warmup 1 --no-cpu-baseline > $O/q_bench_C1S_nested.log 2>&1; tail -1 $O/q_bench_C1S_nested.log | cut -c1-160
timeout -k 10 300 python bench.py --config C5SB --steps 2 --warmup 1 --no-cpu-baseline > $O/q_bench_C5SB.log 2>&1; tail -1 $O/q_bench_C5SB.log | cut -c1-160
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/q_bench_C3.log 2>&1; tail -1 $O/q_bench_C3.log | cut -c1-160
