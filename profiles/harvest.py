"""Copies the summaries a collect_rNN.sh run left under gpurun_out/<dir> into profiles/ (tracked): bench lines, rocprofv3 kernel stats, PMC summaries.

    python profiles/harvest.py r04c r04
"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = os.path.join(ROOT, "gpurun_out", sys.argv[1]), sys.argv[2]
P = os.path.join(ROOT, "profiles")


def last_json(path):
    try:
        lines = [l for l in open(path, errors="replace") if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


for log, name in [("bench.log", "bench_line"), ("bench_prof_c3.log", "bench_line_under_rocprof"), ("bench_c4.log", "bench_line_c4")] + \
                 [("bench_%s.log" % c, "bench_line_%s" % c.lower()) for c in ("C1", "C1L", "C1S", "C2", "C3M", "C5", "C5S", "C5SM", "C5SB")]:
    j = last_json(os.path.join(src, log))
    if j:
        json.dump(j, open(os.path.join(P, "%s_%s.json" % (tag, name)), "w"), indent=1)
        print(name, j["value"], j["roofline"]["frac"], j["roofline"]["avg_launch_ms"])
for d, name in [("prof_c3", "bench_c3"), ("prof_c4", "bench_c4"), ("prof_c1l", "bench_c1l"), ("prof_c5s", "bench_c5s"), ("prof_c5sm", "bench_c5sm"), ("prof_c3m", "bench_c3m"), ("prof_c1s", "bench_c1s")]:
    hits = glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, name)))
for f in ("pmc_summary_c3.json", "pmc_latency_c3.json", "pmc_summary_c4.json", "pmc_summary_c1l.json", "pmc_summary_c5s.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(P, "%s_%s" % (tag, f)))
if os.path.exists(os.path.join(src, "traffic_c3.json")):
    shutil.copy(os.path.join(src, "traffic_c3.json"), os.path.join(P, "traffic_bytes_per_launch.json"))
