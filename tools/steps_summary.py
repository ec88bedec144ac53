"""Condense the step-level PMC passes of tools/profile_round.sh into one row per kernel.

    python tools/steps_summary.py profiles/r05 > profiles/r05_steps_per_kernel_summary.json

Reads <prefix>_steps_pmc_{fetch,write,sq}.csv (tools/prof_summarize.py pmc ... grid) and <prefix>_bench_kernel_stats.csv.
FETCH_SIZE is doubled (gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md: the counter counts 32-byte units where the
documentation says 64); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import csv
import json
import sys


def table(path):
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            out.setdefault(row["kernel"], {})[row["counter"]] = float(row["mean_per_dispatch"])
    return out


def main(prefix):
    fetch, write, sq = (table(f"{prefix}_steps_pmc_{k}.csv") for k in ("fetch", "write", "sq"))
    rows = []
    with open(f"{prefix}_bench_kernel_stats.csv", newline="") as f:
        for r in csv.DictReader(f):
            k = r["kernel"]
            if float(r["pct_gpu_time"]) < 0.4:
                continue
            avg = float(r["avg_us"])
            row = {"kernel": k, "calls": int(float(r["calls"])), "avg_us": avg}
            fe = fetch.get(k, {}).get("FETCH_SIZE")
            wr = write.get(k, {}).get("WRITE_SIZE")
            if fe is not None and wr is not None:
                fmb, wmb = 2.0 * fe * 1024 / 1e6, wr * 1024 / 1e6          # counters are in KB
                row.update(hbm_fetch_MB=round(fmb, 1), hbm_write_MB=round(wmb, 1), TB_per_s=round((fmb + wmb) / avg, 2))
            s = sq.get(k)
            if s and s.get("GRBM_GUI_ACTIVE"):
                row["mfma_busy_pct"] = round(100.0 * s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (s["GRBM_GUI_ACTIVE"] / 8 * 1024), 1)
                if s.get("SQ_LDS_IDX_ACTIVE"):
                    row["lds_conflict_pct_of_lds_active"] = round(100.0 * s.get("SQ_LDS_BANK_CONFLICT", 0.0) / s["SQ_LDS_IDX_ACTIVE"], 1)
            rows.append(row)
    json.dump({"source": f"{prefix}_steps_pmc_{{fetch,write,sq}}.csv + {prefix}_bench_kernel_stats.csv (eager launches of the timed "
               "steps, 8 swaps/step); FETCH_SIZE x2 (gfx950 correction of the guide); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
               "(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)", "rows": rows}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
