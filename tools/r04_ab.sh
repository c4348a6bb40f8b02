# A/B of side-by-side libraries (MTSAMD_LIB: no id check) on the probe sizes.  usage: bash tools/r04_ab.sh tag lib1.so lib2.so ...
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
TAG=$1; shift
for cfg in "C3 512 512 256" "C4 1024 1024 256"; do
  python tests/gpu_ab_cfg.py $cfg "$@" 2>&1 | tee -a $O/ab_$TAG.log
  python tests/gpu_ab_cfg.py $cfg "$@" 2>&1 | tee -a $O/ab_$TAG.log
done
