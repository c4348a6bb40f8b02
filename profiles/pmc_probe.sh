# diagnostic: PMC passes on a 64-spp render of the metric scene (separate passes per counter group)
set -e
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/pp_*
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pp_$tag -o p -- python3 $R/bench.py --spp 64 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pp_$tag.log 2>&1
done
cd $R && python3 tests/pmc_summary.py gpurun_out/pp_FETCH_SIZE gpurun_out/pp_WRITE_SIZE gpurun_out/pp_SQ_INSTS_VALU gpurun_out/pp_SQ_THREAD_CYCLES_VALU
