set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03/gputests_${1:-a}.log 2>&1 || { tail -40 gpurun_out/r03/gputests_${1:-a}.log; exit 1; }
tail -3 gpurun_out/r03/gputests_${1:-a}.log
