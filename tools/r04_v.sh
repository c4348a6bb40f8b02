# volpathmis on lean unit a with 768 threads for its 512 paths (experiment library ab/mis768.so) against the product
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for lib in ab/mis768.so eradiate-kernel_amd/libmtsamd.so ab/mis768.so eradiate-kernel_amd/libmtsamd.so; do
  echo -n "C3M $lib: "; MTSAMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --config C3M --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120
done 2>&1 | tee $O/v_mis768.log
