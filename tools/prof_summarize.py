"""Condense rocprofv3 CSV output into small summaries that fit in profiles/.

  python tools/prof_summarize.py trace  <kernel_trace.csv>        > kernel_stats.csv
  python tools/prof_summarize.py pmc    <counter_collection.csv>  > counters.csv

`trace`: per kernel name (and, for the conv kernel, per grid size = per layer shape): calls, total,
average, min, max duration and share of GPU time.  `pmc`: per kernel name x counter: dispatches,
sum and per-dispatch mean of the counter value."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:90]


def trace(path):
    agg = defaultdict(lambda: [0, 0, 10 ** 18, 0])
    total = 0
    with open(path) as fh:
        for row in csv.DictReader(fh):
            d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            key = short(row["Kernel_Name"])
            if "conv_mfma" in key or "upconv" in key or "conv_bwd" in key or "conv_bf16x3" in key or "conv_c32" in key:
                key += " grid=%d" % (int(row["Grid_Size_X"]) // int(row["Workgroup_Size_X"]))
            a = agg[key]
            a[0] += 1
            a[1] += d
            a[2] = min(a[2], d)
            a[3] = max(a[3], d)
            total += d
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_gpu_time"])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, a[0], "%.1f" % (a[1] / 1e3), "%.2f" % (a[1] / a[0] / 1e3), "%.2f" % (a[2] / 1e3),
                    "%.2f" % (a[3] / 1e3), "%.2f" % (100.0 * a[1] / max(total, 1))])


def pmc(path):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as fh:
        for row in csv.DictReader(fh):
            key = (short(row["Kernel_Name"]), row["Counter_Name"])
            a = agg[key]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "counter", "dispatches", "sum", "mean_per_dispatch"])
    for (k, c), a in sorted(agg.items()):
        w.writerow([k, c, a[0], "%.6g" % a[1], "%.6g" % (a[1] / a[0])])


if __name__ == "__main__":
    {"trace": trace, "pmc": pmc}[sys.argv[1]](sys.argv[2])
