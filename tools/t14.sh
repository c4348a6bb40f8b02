set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(MTSAMD_LPT_DEBUG=1 timeout -k 10 300 python tests/gpu_variants.py 512 512 1024 wga1024) > gpurun_out/r03/costs2.log 2>&1; grep -v "^$" gpurun_out/r03/costs2.log | tail -6
