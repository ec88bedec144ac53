#!/bin/bash
# Run on the GPU box (gpurun): tools/bin/mfma_peak over operand / occupancy variants with rocm-smi sclk + power sampled during each run.
# Output: gpurun_out/mfma_peak.jsonl (one line per variant, with the sampled clocks / power attached).  VERDICT r3 item 3a.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out tools/bin
[ -x tools/bin/mfma_peak ] || hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_peak tools/src/mfma_peak.hip || exit 1
OUT=gpurun_out/mfma_peak.jsonl
: > $OUT
sample() {  # $1 = file; samples until the file $1.stop exists
  while [ ! -e "$1.stop" ]; do
    rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n' >> "$1"; echo >> "$1"
    sleep 0.25
  done
}
for ops in random zero; do
 for wps in 1 2 4; do
  for nacc in 4 8; do
    [ $wps -eq 4 ] && [ $nacc -eq 8 ] && continue          # 4 waves/SIMD x 128 accumulator registers does not fit
    S=/tmp/smi_${ops}_${wps}_${nacc}.jsonl; rm -f $S $S.stop
    sample $S & SP=$!
    LINE=$(timeout 120 tools/bin/mfma_peak $ops $wps $nacc 20000 600)
    touch $S.stop; wait $SP
    python3 - "$S" "$LINE" >> $OUT <<'PY'
import json, sys, re
rows = []
for ln in open(sys.argv[1]):
    try:
        d = json.loads(ln)
    except Exception:
        continue
    c = d.get("card0", {})
    def num(k):
        for kk, v in c.items():
            if k in kk.lower():
                m = re.search(r"([0-9.]+)", str(v))
                if m:
                    return float(m.group(1))
    rows.append((num("sclk"), num("power")))
line = json.loads(sys.argv[2]) if sys.argv[2].startswith("{") else {"error": sys.argv[2]}
mid = rows[len(rows) // 4: max(len(rows) // 4 + 1, 3 * len(rows) // 4)]          # the middle half of the run
sc = [r[0] for r in mid if r[0]]
pw = [r[1] for r in mid if r[1]]
line["smi_samples"] = len(rows)
line["smi_sclk_mhz_mid"] = [min(sc), sum(sc) / len(sc), max(sc)] if sc else None
line["smi_power_w_mid"] = [min(pw), sum(pw) / len(pw), max(pw)] if pw else None
print(json.dumps(line))
PY
  done
 done
done
cat $OUT
rocm-smi --showclocks --showpower --json 2>/dev/null | head -c 1500
