cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
for i in 1 2; do python tests/gpu_ab_cfg.py C4 1024 1024 256 eradiate-kernel_amd/libmtsamd.so 2>&1 | tail -1 | tee -a $O/tiles_c4.log; done
python tests/gpu_ab_cfg.py C4 1024 1024 1024 eradiate-kernel_amd/libmtsamd.so 2>&1 | tail -1 | tee -a $O/tiles_c4.log
