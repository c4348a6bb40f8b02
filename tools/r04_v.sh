# `path` kernels with scene traits (experiment library ab/path_lean.so, -DMTS_TRAITS=110) against the product
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for c in C1L C1 C1S; do
  for lib in ab/path_lean.so eradiate-kernel_amd/libmtsamd.so; do
    echo -n "$c $lib: "; MTSAMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120
  done
done 2>&1 | tee $O/v_path_lean.log
