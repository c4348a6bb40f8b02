set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_kernel_formulation or test_against_oracle or golden" > gpurun_out/r03/form2.log 2>&1 || { tail -30 gpurun_out/r03/form2.log; exit 1; }
tail -3 gpurun_out/r03/form2.log
timeout -k 10 300 python tests/gpu_variants.py 512 512 256 wga1024 wgl1024 > gpurun_out/r03/ab2_c3.log 2>&1; cat gpurun_out/r03/ab2_c3.log
MTSAMD_AB_SCENE=C4 timeout -k 10 300 python tests/gpu_variants.py 1024 1024 64 wga1024 wgl1024 > gpurun_out/r03/ab2_c4.log 2>&1; cat gpurun_out/r03/ab2_c4.log
