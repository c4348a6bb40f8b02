cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
echo "C3M 512 threads: $(python bench.py --config C3M --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110)" | tee -a $O/mis_768.log
echo "C3M 768 threads: $(MTSAMD_WG_THREADS=768 python bench.py --config C3M --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110)" | tee -a $O/mis_768.log
