# the wavefront-split film slots and the pass-order tests
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O



timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "wavefront or passes" > $O/r_tests.log 2>&1 || { tail -30 $O/r_tests.log; exit 1; }
tail -2 $O/r_tests.log
