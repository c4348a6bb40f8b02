cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
bash tools/gputests.sh p
for i in 1 2; do python tests/gpu_ab_cfg.py C2 512 512 256 ab/libmtsamd_prev.so eradiate-kernel_amd/libmtsamd.so 2>&1 | tee -a $O/ab_homog.log; done
python tests/gpu_ab_cfg.py C3 512 512 256 ab/libmtsamd_prev.so eradiate-kernel_amd/libmtsamd.so 2>&1 | tee -a $O/ab_homog.log
