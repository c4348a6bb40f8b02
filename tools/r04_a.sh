# round 4, first GPU call: parity suite on the glibc-exact pmath, bench line, A/B against the correctly rounded build (ab/libmtsamd_cr.so)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
bash tools/gputests.sh a
python bench.py > $O/bench_a.log 2>&1 || { tail -20 $O/bench_a.log; exit 1; }
tail -1 $O/bench_a.log | cut -c1-300
for cfg in "C3 512 512 256" "C4 1024 1024 256" "C2 512 512 256"; do
  python tests/gpu_ab_cfg.py $cfg eradiate-kernel_amd/libmtsamd.so ab/libmtsamd_cr.so 2>&1 | tee -a $O/ab_pmath.log
done
