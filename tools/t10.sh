set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
MTSAMD_AB_SCENE=C4 timeout -k 10 600 python tests/gpu_variants.py 1024 1024 256 wga1024 wga512 wga256 > gpurun_out/r03/wg_c4.log 2>&1; cat gpurun_out/r03/wg_c4.log
