set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r03/t17.log 2>&1 || { tail -40 gpurun_out/r03/t17.log; exit 1; }
tail -3 gpurun_out/r03/t17.log
