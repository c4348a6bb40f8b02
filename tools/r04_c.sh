# round 4, GPU call c: probes of an A/B library (no id check): C3 / C4 / C2 / C3M, workgroup sizes on C4 with expensive-blocks-first
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
LIBS="$@"
for cfg in "C3 512 512 256" "C4 1024 1024 256" "C2 512 512 256"; do
  python tests/gpu_ab_cfg.py $cfg $LIBS 2>&1 | tee -a $O/ab_c.log
done
for kv in wga512 wga256; do
  echo "MTSAMD_KERNEL=$kv" | tee -a $O/ab_c.log
  MTSAMD_KERNEL=$kv python tests/gpu_ab_cfg.py C4 1024 1024 256 $LIBS 2>&1 | tee -a $O/ab_c.log
  MTSAMD_KERNEL=$kv python tests/gpu_ab_cfg.py C3 512 512 256 $LIBS 2>&1 | tee -a $O/ab_c.log
done
