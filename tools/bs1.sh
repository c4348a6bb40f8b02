set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export MTSAMD_LIB=$GRAFT_REPO_ROOT/ab/bs.so
for k in wga1024 wgl1024; do
  for c in C3 C4; do
    echo "== $k $c"; MTSAMD_KERNEL=$k timeout -k 10 300 python tests/gpu_blockstats.py 512 512 64 $c
  done
done > gpurun_out/r03/bs1.log 2>&1
cat gpurun_out/r03/bs1.log
