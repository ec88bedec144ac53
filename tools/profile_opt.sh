#!/bin/bash
# Run on the GPU box through gpurun: rocprofv3 kernel trace of the config-3 optimisation leg (bench.py --opt-only),
# condensed per kernel into gpurun_out/profiles/<tag>_opt_kernel_stats.csv.
set -u
TAG=${1:-r02}
STEPS=${2:-20}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles
RAW=/tmp/e4s_prof_opt
mkdir -p "$OUT" "$RAW"
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $RAW/opt -o opt -- python bench.py --opt-only --opt-steps $STEPS > $RAW/opt.log 2>&1 < /dev/null
echo "rc=$?"
t=$(find $RAW/opt -name "*_kernel_trace.csv" | head -1)
if [ -n "$t" ]; then python tools/prof_summarize.py trace "$t" > $OUT/${TAG}_opt_kernel_stats.csv; head -50 $OUT/${TAG}_opt_kernel_stats.csv | cut -c1-160; fi
grep -h '^{' $RAW/opt.log > $OUT/${TAG}_opt_line.json
