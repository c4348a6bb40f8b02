set -e
cd $GRAFT_REPO_ROOT
R=${ROUND:-r04}
mkdir -p gpurun_out/$R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/$R/gputests_${1:-a}.log 2>&1 || { tail -40 gpurun_out/$R/gputests_${1:-a}.log; exit 1; }
tail -3 gpurun_out/$R/gputests_${1:-a}.log
