set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 500 python bench.py --config C5SM --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_c5sm.log 2>&1; tail -1 gpurun_out/r03/bench_c5sm.log
MTSAMD_KERNEL=nested timeout -k 10 500 python bench.py --config C5SM --spp 16 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03/bench_c5sm_nested.log 2>&1; tail -1 gpurun_out/r03/bench_c5sm_nested.log
timeout -k 10 500 python bench.py --config C5S --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_c5s.log 2>&1; tail -1 gpurun_out/r03/bench_c5s.log
