set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 300 python tests/gpu_passes.py C3 512 512 1024 > gpurun_out/r03/passes_c3.log 2>&1; cat gpurun_out/r03/passes_c3.log
timeout -k 10 300 python tests/gpu_passes.py C4 1024 1024 256 > gpurun_out/r03/passes_c4.log 2>&1; cat gpurun_out/r03/passes_c4.log
