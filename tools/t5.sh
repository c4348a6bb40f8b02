set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stopped or cancel or every_kernel" > gpurun_out/r03/t5.log 2>&1 || { tail -40 gpurun_out/r03/t5.log; exit 1; }
tail -3 gpurun_out/r03/t5.log
