# Round-2 evidence run on the GPU box: GPU tests, smoke, bench (C3 metric), rocprofv3 kernel stats, PMC traffic + issue counters for
# C3, kernel stats + PMC for C4.  Usage: gpurun -- 'bash profiles/collect_r02.sh [tag]'
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python bench.py > $O/bench.log 2>&1 || { tail -20 $O/bench.log; exit 1; }
tail -1 $O/bench.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o p -- python3 $R/bench.py --no-cpu-baseline > $O/bench_prof_c3.log 2>&1
tail -1 $O/bench_prof_c3.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write.log 2>&1
for grp in "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $O/pp_$tag -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pp_$tag.log 2>&1
done
cd $R
python3 tests/pmc_summary.py --json $O/pmc_summary_c3.json --probe "bench.py --steps 1 (C3 512x512x1024)" $O/pmc_fetch $O/pmc_write $O/pp_SQ_INSTS_VALU $O/pp_SQ_THREAD_CYCLES_VALU
python3 profiles/collect_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic_c3.json $O/pmc_fetch.log $O/pmc_summary_c3.json
# C4 (the Eradiate atmosphere): side measurement with its own kernel stats and counters
python bench.py --config C4 --steps 2 --warmup 1 > $O/bench_c4.log 2>&1; tail -1 $O/bench_c4.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python3 $R/bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_prof_c4.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $O/c4_$tag -o p -- python3 $R/bench.py --config C4 --spp 256 --steps 1 --warmup 0 --no-cpu-baseline > $O/c4_$tag.log 2>&1
done
cd $R
python3 tests/pmc_summary.py --json $O/pmc_summary_c4.json --probe "bench.py --config C4 --spp 256 --steps 1 (1024x1024x256)" $O/c4_FETCH_SIZE $O/c4_WRITE_SIZE $O/c4_SQ_INSTS_VALU $O/c4_SQ_THREAD_CYCLES_VALU
for c in C1 C1L C2 C3M C5S; do python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$c.log 2>&1; tail -1 $O/bench_$c.log; done
# C5 = 16 wavelength batches of the C4 atmosphere: 256 spp per batch here (the full 4096 spp take ~5 minutes per step)
python bench.py --config C5 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_C5.log 2>&1; tail -1 $O/bench_C5.log
find $O -name "*kernel_stats*" | head
