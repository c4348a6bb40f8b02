cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
export MTSAMD_LIB=$PWD/ab/libmtsamd_bs.so
echo "== cost-sorted tiles (default)" | tee -a $O/blockstats_tiles.log
MTSAMD_LPT_DEBUG=1 python tests/gpu_blockstats.py 1024 1024 256 C4 2>&1 | tail -14 | tee -a $O/blockstats_tiles.log
echo "== MTSAMD_LPT=0 (spatial blocks, spiral order)" | tee -a $O/blockstats_tiles.log
MTSAMD_LPT=0 python tests/gpu_blockstats.py 1024 1024 256 C4 2>&1 | tail -5 | tee -a $O/blockstats_tiles.log
