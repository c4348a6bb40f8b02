cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
MTSAMD_LPT_DEBUG=1 python tests/gpu_ab_cfg.py C4 1024 1024 256 eradiate-kernel_amd/libmtsamd.so 2>&1 | tail -14 | tee -a $O/tiles_c4.log
MTSAMD_LPT=0 python tests/gpu_ab_cfg.py C4 1024 1024 256 eradiate-kernel_amd/libmtsamd.so 2>&1 | tail -2 | tee -a $O/tiles_c4.log
python tests/gpu_ab_cfg.py C3 512 512 256 eradiate-kernel_amd/libmtsamd.so 2>&1 | tail -2 | tee -a $O/tiles_c4.log
bash tools/gputests.sh g
