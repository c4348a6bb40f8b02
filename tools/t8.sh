set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or against_oracle or every_kernel or volpathmis or stopped or cancel or spectral_variant" > gpurun_out/r03/t8.log 2>&1 || { tail -40 gpurun_out/r03/t8.log; exit 1; }
tail -3 gpurun_out/r03/t8.log
timeout -k 10 300 python tests/gpu_variants.py 512 512 256 wga1024 > gpurun_out/r03/ab8_c3.log 2>&1; cat gpurun_out/r03/ab8_c3.log
MTSAMD_AB_SCENE=C4 timeout -k 10 300 python tests/gpu_variants.py 1024 1024 64 wga1024 > gpurun_out/r03/ab8_c4.log 2>&1; cat gpurun_out/r03/ab8_c4.log
MTSAMD_AB_INTEGRATOR=volpathmis timeout -k 10 300 python tests/gpu_variants.py 512 512 256 wga512 > gpurun_out/r03/ab8_c3m.log 2>&1; cat gpurun_out/r03/ab8_c3m.log
