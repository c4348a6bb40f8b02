# closing checks: the wide fuzz soak and the GPU suite with the final binary
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python tests/gpu_fuzz_soak.py 0 2000 2>&1 | tail -3 | tee $O/fuzz_soak_final.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gputests_final4.log 2>&1 || { tail -30 $O/gputests_final4.log; exit 1; }
tail -2 $O/gputests_final4.log
