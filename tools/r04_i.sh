cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
for cfg in C4 C5S C5SM C5; do
  for m in 1 x; do
    if [ $m = x ]; then unset MTSAMD_LPT; else export MTSAMD_LPT=$m; fi
    echo "$cfg MTSAMD_LPT=$m: $(python bench.py --config $cfg --spp 256 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120)" | tee -a $O/tiles_ab.log
  done
done
