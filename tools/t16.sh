set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wavefront or expensive_blocks or golden or against_oracle" > gpurun_out/r03/t16.log 2>&1 || { tail -50 gpurun_out/r03/t16.log; exit 1; }
tail -3 gpurun_out/r03/t16.log
for c in C1 C1W; do timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-150; done
