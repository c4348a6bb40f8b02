cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
bash tools/gputests.sh n
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-110 | tee -a $O/bench_n.log
python - <<'PY' 2>&1 | tee -a $O/wavefront_ring.log
import importlib, os, time
pkg = importlib.import_module("eradiate-kernel_amd"); scenes = importlib.import_module("eradiate-kernel_amd.scenes")
pkg.set_variant("gpu_rgb")
for kern in ("default", "flat"):
    if kern == "flat": os.environ["MTSAMD_KERNEL"] = "flat"
    d = scenes.c3_heterogeneous(256, 256, 256); d["sensor"]["sampler"]["wavefront"] = True
    sc = pkg.load_dict(d); best = 1e9
    for r in range(3):
        sc.integrator().render(sc, sc.sensors()[0]); st = sc.integrator().last_stats; best = min(best, st["kernel_ms"])
    print("C3 256x256x256 wavefront streams, MTSAMD_KERNEL=%s: variant %d, %.1f ms -> %.1f Msamples/s" % (kern, st["kernel_variant"], best, st["samples"] / best / 1e3))
PY
