set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "independent or chromatic" > gpurun_out/r03/t19.log 2>&1 || { tail -40 gpurun_out/r03/t19.log; exit 1; }
grep "channel\|passed" gpurun_out/r03/t19.log | tail -8
