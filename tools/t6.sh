set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03/gputests_b.log 2>&1 || { tail -60 gpurun_out/r03/gputests_b.log; exit 1; }
tail -3 gpurun_out/r03/gputests_b.log
timeout -k 10 500 python bench.py --config C5SB --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_c5sb.log 2>&1; tail -1 gpurun_out/r03/bench_c5sb.log | cut -c1-300
MTSAMD_KERNEL=nested timeout -k 10 500 python bench.py --config C5SB --spp 16 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03/bench_c5sb_nested.log 2>&1; tail -1 gpurun_out/r03/bench_c5sb_nested.log | cut -c1-300
