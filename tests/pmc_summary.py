"""Summarise rocprofv3 --pmc CSV output for the render kernels (diagnostic helper)."""
import csv, collections, sys
for d in sys.argv[1:]:
    rows = list(csv.DictReader(open(d + "/p_counter_collection.csv")))
    agg = collections.defaultdict(float); t = 0
    for r in rows:
        if "render_kernel" in r["Kernel_Name"] and "<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(d, "ms %.1f" % t, {k: "%.4g" % v for k, v in sorted(agg.items())})
