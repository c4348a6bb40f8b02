# round 4, GPU call b: full parity suite (new tests: SIGINT, lost path, C5 batch), bench line, probes
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
bash tools/gputests.sh ${1:-b}
python bench.py > $O/bench_${1:-b}.log 2>&1 || { tail -20 $O/bench_${1:-b}.log; exit 1; }
tail -1 $O/bench_${1:-b}.log | cut -c1-400
for cfg in "C3 512 512 256" "C4 1024 1024 256" "C2 512 512 256"; do
  python tests/gpu_ab_cfg.py $cfg eradiate-kernel_amd/libmtsamd.so 2>&1 | tee -a $O/ab_${1:-b}.log
done
