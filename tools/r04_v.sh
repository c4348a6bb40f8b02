# lean unit c (BVH allowed): parity cases, then a canopy under the C4 atmosphere, lean against general
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "lean" > $O/v_tests.log 2>&1 || { tail -30 $O/v_tests.log; exit 1; }
tail -2 $O/v_tests.log
timeout -k 10 300 python tests/gpu_canopy_ab.py 2>&1 | tee $O/v_canopy.log
