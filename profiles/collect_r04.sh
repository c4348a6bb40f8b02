# Round-4 evidence run on the GPU box: bench lines of every configuration, rocprofv3 kernel stats (C3, C4, C1L), PMC traffic + issue +
# latency / LDS counters for C3, issue counters for C4 and C1L.  Usage: gpurun -- 'bash profiles/collect_r04.sh [part]'   (part: a | b | c | d)
set -e
PART=${1:-a}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$R/gpurun_out/r04c; mkdir -p $O
cd $R
if [ $PART = a ]; then
python bench.py > $O/bench.log 2>&1 || { tail -20 $O/bench.log; exit 1; }
tail -1 $O/bench.log | cut -c1-200
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o p -- python3 $R/bench.py --no-cpu-baseline > $O/bench_prof_c3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write.log 2>&1
for grp in "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $O/pp_$tag -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pp_$tag.log 2>&1
done
i=0
for grp in "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/lat$i -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/lat$i.log 2>&1 || { tail -5 $O/lat$i.log; }
done
cd $R
python3 tests/pmc_summary.py --json $O/pmc_summary_c3.json --probe "bench.py --steps 1 (C3 512x512x1024)" $O/pmc_fetch $O/pmc_write $O/pp_SQ_INSTS_VALU $O/pp_SQ_THREAD_CYCLES_VALU > /dev/null
python3 tests/pmc_summary.py --json $O/pmc_latency_c3.json --probe "bench.py --steps 1 (C3 512x512x1024), latency counters" $O/lat1 $O/lat2 $O/lat3 > /dev/null
python3 profiles/collect_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic_c3.json $O/pmc_fetch.log $O/pmc_summary_c3.json | cut -c1-300
fi
if [ $PART = b ]; then
# C4 (the Eradiate atmosphere): bench line, kernel stats, counters (256-spp probe)
python bench.py --config C4 --steps 2 --warmup 1 > $O/bench_c4.log 2>&1; tail -1 $O/bench_c4.log | cut -c1-200
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python3 $R/bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_prof_c4.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $O/c4_$tag -o p -- python3 $R/bench.py --config C4 --spp 256 --steps 1 --warmup 0 --no-cpu-baseline > $O/c4_$tag.log 2>&1
done
cd $R
python3 tests/pmc_summary.py --json $O/pmc_summary_c4.json --probe "bench.py --config C4 --spp 256 --steps 1 (1024x1024x256)" $O/c4_FETCH_SIZE $O/c4_WRITE_SIZE $O/c4_SQ_INSTS_VALU $O/c4_SQ_THREAD_CYCLES_VALU > /dev/null
cat $O/pmc_summary_c4.json | tail -12
fi
if [ $PART = c ]; then
for c in C1 C1L C1S C2 C3M C5S C5SM C5SB; do python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$c.log 2>&1; tail -1 $O/bench_$c.log | cut -c1-160; done
# C5 = 16 wavelength batches of the C4 atmosphere: 256 spp per batch here (the full 4096 spp take minutes per step)
python bench.py --config C5 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_C5.log 2>&1; tail -1 $O/bench_C5.log | cut -c1-160
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c1l -o p -- python3 $R/bench.py --config C1L --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_prof_c1l.log 2>&1
for grp in "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $O/c1l_$tag -o p -- python3 $R/bench.py --config C1L --steps 1 --warmup 0 --no-cpu-baseline > $O/c1l_$tag.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5sm -o p -- python3 $R/bench.py --config C5SM --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_prof_c5sm.log 2>&1
cd $R
python3 tests/pmc_summary.py --json $O/pmc_summary_c1l.json --probe "bench.py --config C1L --steps 1 (512x512x256, path)" $O/c1l_SQ_INSTS_VALU $O/c1l_SQ_THREAD_CYCLES_VALU > /dev/null
cat $O/pmc_summary_c1l.json | tail -10
fi
if [ $PART = d ]; then
# C5S (the spectral variant on the C4 atmosphere): kernel stats and issue counters (128-spp probe)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5s -o p -- python3 $R/bench.py --config C5S --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_prof_c5s.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $O/c5s_$tag -o p -- python3 $R/bench.py --config C5S --spp 128 --steps 1 --warmup 0 --no-cpu-baseline > $O/c5s_$tag.log 2>&1
done
cd $R
python3 tests/pmc_summary.py --json $O/pmc_summary_c5s.json --probe "bench.py --config C5S --spp 128 --steps 1 (1024x1024x128, gpu_spectral)" $O/c5s_FETCH_SIZE $O/c5s_WRITE_SIZE $O/c5s_SQ_INSTS_VALU $O/c5s_SQ_THREAD_CYCLES_VALU > /dev/null
cat $O/pmc_summary_c5s.json | tail -12
fi
find $O -name "*kernel_stats*" | head
if [ $PART = e ]; then
# kernel stats of the kernels that changed late in the round: volpathmis on the metric scene, `path` in the spectral variant
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3m -o p -- python3 $R/bench.py --config C3M --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_prof_c3m.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c1s -o p -- python3 $R/bench.py --config C1S --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_prof_c1s.log 2>&1
cd $R
tail -1 $O/bench_prof_c3m.log | cut -c1-160; tail -1 $O/bench_prof_c1s.log | cut -c1-160
fi
