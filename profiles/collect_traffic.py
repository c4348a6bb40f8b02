"""Reduce the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass on gfx950)
of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline [...]` into a traffic record.

usage: python profiles/collect_traffic.py <fetch_pass_dir> <write_pass_dir> <out.json> <bench_log_of_one_pass> [pmc_summary.json]

The record carries the configuration it describes (`run`: config, size, spp, kernel -- read from the JSON line the profiled
bench.py printed); bench.py attaches `roofline.traffic` only to runs of exactly that configuration.

Units / corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are reported in KiB
summed over the 16 TCC channels x 8 XCDs; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so
reads are doubled; WRITE_SIZE is exact.  The guide calibrates the doubling on 16-byte-per-lane streaming
reads; this kernel issues 4-byte gathers, so 2 x FETCH is kept as the (conservative) figure and the raw
value is recorded next to it."""
import csv, glob, json, os, re, sys


kernel_name = None


def kernel_sum(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    global kernel_name
    tot, n = 0.0, set()
    for path in f:
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "render_kernel" in k and "<false" in k and r["Counter_Name"] == counter:   # the timed (non-counting) kernel
                tot += float(r["Counter_Value"]); n.add(r["Dispatch_Id"]); kernel_name = k.split("(")[0].replace("void mtsamd::", "")
    return tot, max(len(n), 1)


fetch_kib, nf = kernel_sum(sys.argv[1], "FETCH_SIZE")
write_kib, nw = kernel_sum(sys.argv[2], "WRITE_SIZE")
line = json.loads([l for l in open(sys.argv[4]) if l.startswith("{")][-1])
m = re.search(r"(\d+)x(\d+)x(\d+)spp", line["config"]["workload"])
res = re.search(r"(\d+)\^3", line["config"]["workload"])
cfg = re.match(r"(C\d)", line["config"]["workload"]).group(1)
out = {"run": {"config": cfg, "width": int(m.group(1)), "height": int(m.group(2)), "spp": int(m.group(3)), "res": int(res.group(1)) if res else 128,
               "n_gpus": line["n_gpus"], "kernel": line["roofline"]["kernel"]},
       "kernel": kernel_name, "launches": [nf, nw],
       "fetch_bytes_raw": fetch_kib / nf * 1024.0, "fetch_bytes_corrected": 2.0 * fetch_kib / nf * 1024.0,
       "write_bytes": write_kib / nw * 1024.0}
out["bytes_per_launch"] = out["fetch_bytes_corrected"] + out["write_bytes"]
if len(sys.argv) > 5:
    pmc = json.load(open(sys.argv[5]))
    out["derived"] = pmc["derived"]                     # tests/pmc_summary.py: valu_pipe_busy, wave_wait_share, lane_utilisation ...
    out["derived_source"] = pmc["probe"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
