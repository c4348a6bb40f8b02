# C3 at several grid resolutions: the same statistics on grids that sit in L1 / L2 / beyond -- how much of a step is lookup latency?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for r in 128 64 32 8; do
  python bench.py --spp 256 --res $r --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04/res_$r.log 2>&1
  python - $r <<'PY' | tee -a gpurun_out/r04/res_sweep.log
import json, sys
r = sys.argv[1]
l = json.loads(open("gpurun_out/r04/res_%s.log" % r).read().strip().splitlines()[-1])
rf = l["roofline"]
print("res %s: %.1f Msamples/s, iter %.2f lookup %.2f nee %.2f per sample, model frac %.3f" % (r, l["value"], rf["n_iter_per_sample"], rf["n_lookup_per_sample"], rf["n_nee_step_per_sample"], rf["frac"]))
PY
done
