set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(MTSAMD_LPT=2 MTSAMD_LPT_DEBUG=1 timeout -k 10 300 python tests/gpu_variants.py 512 512 1024 wga1024
 MTSAMD_LPT=2 MTSAMD_LPT_DEBUG=1 MTSAMD_AB_SCENE=C4 timeout -k 10 300 python tests/gpu_variants.py 1024 1024 512 wga1024
 MTSAMD_LPT=2 MTSAMD_LPT_DEBUG=1 MTSAMD_AB_INTEGRATOR=volpathmis timeout -k 10 300 python tests/gpu_variants.py 512 512 512 wga512) > gpurun_out/r03/costs.log 2>&1; grep -v "^$" gpurun_out/r03/costs.log | tail -12
