"""Condense rocprofv3 CSV output into small summaries that fit in profiles/.

  python tools/prof_summarize.py trace  <kernel_trace.csv>        > kernel_stats.csv
  python tools/prof_summarize.py pmc    <counter_collection.csv> [grid]  > counters.csv     (grid: conv kernels keyed per grid size too)

`trace`: per kernel name (and, for the conv kernel, per grid size = per layer shape): calls, total,
average, min, max duration and share of GPU time.  `pmc`: per kernel name x counter: dispatches,
sum and per-dispatch mean of the counter value."""
import csv
import sys
from collections import defaultdict


CONV_KEYS = ("conv_mfma", "upconv", "conv_bwd", "conv_bf16x3", "conv_c32", "conv_region", "wino")


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
            if any(k in key for k in CONV_KEYS):
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


def pmc(path, per_grid=False):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = short(row["Kernel_Name"])
            if per_grid and any(k in name for k in CONV_KEYS):
                gs = row.get("Grid_Size") or row.get("Grid_Size_X")
                wg = row.get("Workgroup_Size") or row.get("Workgroup_Size_X")
                if gs and wg:
                    name += " grid=%d" % (int(gs) // max(int(wg), 1))
            key = (name, row["Counter_Name"])
            a = agg[key]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "counter", "dispatches", "sum", "mean_per_dispatch"])
    for (k, c), a in sorted(agg.items()):
        w.writerow([k, c, a[0], "%.6g" % a[1], "%.6g" % (a[1] / a[0])])


if __name__ == "__main__":
    if sys.argv[1] == "pmc":
        pmc(sys.argv[2], len(sys.argv) > 3 and sys.argv[3] == "grid")
    else:
        trace(sys.argv[2])
