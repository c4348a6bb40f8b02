# diagnostic: instruction-cache counters of a 64-spp render of the metric scene
set -e
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/pp_ic
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pp_ic -o p -- python3 $R/bench.py --spp 64 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pp_ic.log 2>&1
cd $R && python3 tests/pmc_summary.py gpurun_out/pp_ic
