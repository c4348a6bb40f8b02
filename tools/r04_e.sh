cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
for cfg in "1024 1024 256 C4" "512 512 256 C3"; do
  MTSAMD_LIB=$PWD/ab/libmtsamd_bs.so python tests/gpu_blockstats.py $cfg 2>&1 | tail -4 | tee -a $O/blockstats_pop.log
done
bash tools/gputests.sh e
