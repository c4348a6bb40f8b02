# blockstats of C4 at two sample counts (is the idle share a tail or a steady state?), and the spectral MIS kernel served by 512 threads
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
for spp in 256 2048; do
  MTSAMD_LIB=$PWD/ab/libmtsamd_bs.so python tests/gpu_blockstats.py 1024 1024 $spp C4 2>&1 | tee -a $O/blockstats_c4.log
done
for nt in 0 512; do
  echo "MTSAMD_WG_THREADS=$nt" | tee -a $O/mis512.log
  if [ $nt = 0 ]; then python bench.py --config C5SM --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee -a $O/mis512.log
  else MTSAMD_WG_THREADS=$nt python bench.py --config C5SM --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee -a $O/mis512.log; fi
done
