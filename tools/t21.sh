set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_kernel_formulation or stopped" > gpurun_out/r03/t21.log 2>&1 || { tail -30 gpurun_out/r03/t21.log; exit 1; }
tail -2 gpurun_out/r03/t21.log
timeout -k 10 300 python tests/gpu_variants.py 512 512 256 wga1024 wgl1024 > gpurun_out/r03/ab21_c3.log 2>&1; cat gpurun_out/r03/ab21_c3.log
MTSAMD_LPT=0 MTSAMD_AB_SCENE=C4 timeout -k 10 300 python tests/gpu_variants.py 1024 1024 64 wga1024 wgl1024 > gpurun_out/r03/ab21_c4.log 2>&1; cat gpurun_out/r03/ab21_c4.log
if [ -f ab/bs.so ]; then MTSAMD_LIB=$GRAFT_REPO_ROOT/ab/bs.so MTSAMD_KERNEL=wgl1024 timeout -k 10 300 python tests/gpu_blockstats.py 512 512 64 C3 > gpurun_out/r03/bs21.log 2>&1; cat gpurun_out/r03/bs21.log; fi
