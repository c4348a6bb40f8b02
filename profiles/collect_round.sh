set -e
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests8.log 2>&1 || { tail -30 gpurun_out/gputests8.log; exit 1; }
tail -3 gpurun_out/gputests8.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke8.log 2>&1 || { tail -20 gpurun_out/smoke8.log; exit 1; }
tail -2 gpurun_out/smoke8.log
timeout -k 10 600 python bench.py > gpurun_out/bench8.log 2>&1 || { tail -20 gpurun_out/bench8.log; exit 1; }
tail -1 gpurun_out/bench8.log
cd /tmp
rm -rf $R/gpurun_out/prof_r1c $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1c -o p -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_prof8.log 2>&1
tail -1 $R/gpurun_out/bench_prof8.log
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
cd $R && python profiles/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/traffic.json
find gpurun_out/prof_r1c -name "*stats*" | head
