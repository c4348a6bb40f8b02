cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
MTSAMD_LPT=3 timeout -k 10 500 python tests/gpu_fuzz_soak.py 0 120 2>&1 | tail -6 | tee -a $O/fuzz_soak_tiles.log
timeout -k 10 500 python tests/gpu_fuzz_soak.py 120 260 2>&1 | tail -6 | tee -a $O/fuzz_soak_default.log
