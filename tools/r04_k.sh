cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
for lib in eradiate-kernel_amd/libmtsamd.so ab/libmtsamd_pw3.so; do
  for cfg in C1L C1; do
    echo "$cfg $(basename $lib): $(MTSAMD_LIB=$PWD/$lib python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110)" | tee -a $O/path_waves.log
  done
done
