# Where a wave's time goes, beyond profiles/collect_r02.sh: in-flight levels (LEVEL / INSTS = mean latency in cycles), active-instruction
# cycles per type, instruction fetch and LDS conflicts.  Usage: gpurun -- 'bash profiles/pmc_latency.sh [tag]'   (one --pmc group per pass)
set -e
TAG=${1:-r02_latency}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp
i=0
for grp in "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/g$i.log 2>&1 || { tail -5 $O/g$i.log; }
done
cd $R
python3 tests/pmc_summary.py --json $O/latency_c3.json --probe "bench.py --steps 1 (C3 512x512x1024), latency counters" $O/g1 $O/g2 $O/g3
