set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout -k 10 300 python tests/gpu_variants.py 512 512 1024 wga1024 wga512 wga256) > gpurun_out/r03/wg_c3.log 2>&1; cat gpurun_out/r03/wg_c3.log
