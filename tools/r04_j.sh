cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
for kv in wga256 wga512; do
  echo "C5S MTSAMD_KERNEL=$kv: $(MTSAMD_KERNEL=$kv python bench.py --config C5S --spp 256 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110)" | tee -a $O/spectral_wg.log
done
echo "C5SB: $(python bench.py --config C5SB --spp 256 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110)" | tee -a $O/spectral_wg.log
echo "C3M: $(python bench.py --config C3M --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110)" | tee -a $O/spectral_wg.log
echo "C3M LPT=3: $(MTSAMD_LPT=3 python bench.py --config C3M --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110)" | tee -a $O/spectral_wg.log
