set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "golden or c1 or cornell or path or every_kernel or fuzz or distant or stopped or cancel" > gpurun_out/r03/t7.log 2>&1 || { tail -40 gpurun_out/r03/t7.log; exit 1; }
tail -3 gpurun_out/r03/t7.log
for k in default nested; do
  if [ $k = nested ]; then export MTSAMD_KERNEL=nested; fi
  timeout -k 10 300 python bench.py --config C1L --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_c1l_$k.log 2>&1; tail -1 gpurun_out/r03/bench_c1l_$k.log | cut -c1-200
  timeout -k 10 300 python bench.py --config C1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_c1_$k.log 2>&1; tail -1 gpurun_out/r03/bench_c1_$k.log | cut -c1-200
done
