"""Condense rocprofv3 CSV output into small summaries that fit in profiles/.

  python tools/prof_summarize.py trace  <kernel_trace.csv>        > kernel_stats.csv
  python tools/prof_summarize.py pmc    <counter_collection.csv> [grid]  > counters.csv     (grid: conv kernels keyed per grid size too)
  python tools/prof_summarize.py replay <kernel_trace.csv> [info.json]   > kernel_stats.csv (the graph REPLAYS at the end of the trace only)

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


def replay(path, out_json=None):
    """Kernel trace of a run that ends with K replays of ONE captured graph (bench.py --steps-only): keep the replays only.  A replay
    dispatches the same kernel sequence every time, so the trace (sorted by start time) ends periodically: the period n is the number of
    kernels per replay, the number of whole periods at the end is K.  Per kernel: calls / total / average over those K * n dispatches;
    `out_json`: {kernels_per_replay, replays, sum_kernel_ms_per_replay, span_ms_per_replay} -- round 4's table mixed the eager warm-up
    passes and the one-time pack kernels into its "per replay" sums."""
    import json
    rows = []
    with open(path) as fh:
        for row in csv.DictReader(fh):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short(row["Kernel_Name"]),
                         int(row["Grid_Size_X"]) // max(int(row["Workgroup_Size_X"]), 1)))
    rows.sort()
    names = [(r[2], r[3]) for r in rows]
    n = None
    for skip in range(0, 40):                         # a few dispatches may follow the last replay (flag reset, validation)
        tail_names = names[:len(names) - skip] if skip else names
        for cand in range(40, len(tail_names) // 2 + 1):
            if tail_names[-cand:] == tail_names[-2 * cand:-cand]:
                n = cand
                break
        if n is not None:
            if skip:
                rows, names = rows[:len(rows) - skip], tail_names
            break
    if n is None:
        raise SystemExit("no periodic tail found: is this a graph-replay trace?")
    k = 1
    while (k + 1) * n <= len(names) and names[-(k + 1) * n:-k * n] == names[-n:]:
        k += 1
    tail = rows[-k * n:]
    agg = defaultdict(lambda: [0, 0, 10 ** 18, 0])
    total = 0
    for st, en, name, grid in tail:
        key = name + (" grid=%d" % grid if any(c in name for c in CONV_KEYS) else "")
        a = agg[key]
        d = en - st
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
        total += d
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls_per_replay", "total_us_per_replay", "avg_us", "min_us", "max_us", "pct_gpu_time"])
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([key, "%.2f" % (a[0] / k), "%.1f" % (a[1] / 1e3 / k), "%.2f" % (a[1] / a[0] / 1e3), "%.2f" % (a[2] / 1e3),
                    "%.2f" % (a[3] / 1e3), "%.2f" % (100.0 * a[1] / max(total, 1))])
    spans = [(tail[(i + 1) * n - 1][1] - tail[i * n][0]) / 1e6 for i in range(k)]
    info = {"kernels_per_replay": n, "replays": k, "sum_kernel_ms_per_replay": round(total / 1e6 / k, 3),
            "span_ms_per_replay": [round(v, 3) for v in spans]}
    if out_json:
        with open(out_json, "w") as fh:
            json.dump(info, fh)
    sys.stderr.write(json.dumps(info) + "\n")


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
    elif sys.argv[1] == "replay":
        replay(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        trace(sys.argv[2])
