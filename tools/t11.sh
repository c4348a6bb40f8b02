set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or against_oracle or every_kernel or sharded or rehearsal" > gpurun_out/r03/t11.log 2>&1 || { tail -40 gpurun_out/r03/t11.log; exit 1; }
tail -3 gpurun_out/r03/t11.log
for lpt in 0 1; do
  echo "MTSAMD_LPT=$lpt"
  MTSAMD_LPT=$lpt MTSAMD_AB_SCENE=C4 timeout -k 10 300 python tests/gpu_variants.py 1024 1024 256 wga1024 2>&1 | tail -1
  MTSAMD_LPT=$lpt timeout -k 10 300 python tests/gpu_variants.py 1024 1024 256 wga1024 2>&1 | tail -1
done > gpurun_out/r03/lpt.log 2>&1; cat gpurun_out/r03/lpt.log
timeout -k 10 500 python bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_c4_lpt.log 2>&1; tail -1 gpurun_out/r03/bench_c4_lpt.log | cut -c1-250
