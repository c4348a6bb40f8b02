set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for l in pw3 pw5; do
  MTSAMD_LIB=$GRAFT_REPO_ROOT/ab/$l.so timeout -k 10 300 python bench.py --config C1L --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120
done > gpurun_out/r03/pathwaves.log 2>&1; cat gpurun_out/r03/pathwaves.log
timeout -k 10 300 python bench.py --config C1L --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120
