# lean units with their remaining callees inline: the GPU suite, then bench lines
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/t_tests.log 2>&1 || { tail -30 $O/t_tests.log; exit 1; }
tail -2 $O/t_tests.log
for c in C5S C5SM C5SB C2 C3; do timeout -k 10 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/t_bench_$c.log 2>&1; tail -1 $O/t_bench_$c.log | cut -c1-160; done
timeout -k 10 300 python bench.py --config C4 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/t_bench_C4.log 2>&1; tail -1 $O/t_bench_C4.log | cut -c1-160
timeout -k 10 300 python bench.py --config C5 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/t_bench_C5.log 2>&1; tail -1 $O/t_bench_C5.log | cut -c1-160
