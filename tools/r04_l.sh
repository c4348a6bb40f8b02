cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "mis or tiles or stopped or cancel or lost" > $O/gputests_mis.log 2>&1 || { tail -30 $O/gputests_mis.log; exit 1; }
tail -2 $O/gputests_mis.log
for cfg in C5SM C3M; do
  echo "$cfg: $(python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110)" | tee -a $O/mis_park.log
done
echo "C5SM 512 threads: $(MTSAMD_WG_THREADS=512 python bench.py --config C5SM --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110)" | tee -a $O/mis_park.log
echo "C3M wga256: $(MTSAMD_KERNEL=wga256 python bench.py --config C3M --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110)" | tee -a $O/mis_park.log
