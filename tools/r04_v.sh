# the step's exponential evaluated while the grid gathers are in flight (experiment library ab/tr_early.so) against the product, lean units
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for rep in 1 2; do for lib in ab/tr_early.so eradiate-kernel_amd/libmtsamd.so; do
  echo -n "C3 $lib: "; MTSAMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c80-130
  echo -n "C4 $lib: "; MTSAMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --config C4 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c40-100
done; done 2>&1 | tee $O/v_tr_early.log
