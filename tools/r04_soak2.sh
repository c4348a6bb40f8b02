cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python tests/gpu_fuzz_soak.py 0 2000 2>&1 | tail -8 | tee $O/fuzz_soak_wide.log
