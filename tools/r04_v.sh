# cost-sorted tiles (MTSAMD_LPT=3) against whole blocks by cost (default for rgb) on the lean units
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for lpt in 3 1 3 1; do
  echo -n "C4 256 spp MTSAMD_LPT=$lpt: "; MTSAMD_LPT=$lpt timeout -k 10 300 python bench.py --config C4 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110
done 2>&1 | tee $O/v_lpt_lean.log
for lpt in 3 1; do
  echo -n "C3 MTSAMD_LPT=$lpt: "; MTSAMD_LPT=$lpt timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110
done 2>&1 | tee -a $O/v_lpt_lean.log
