#!/bin/bash
# Minimal refresh: rocprofv3 kernel trace of the headline probe and of the timed steps (no PMC passes).
set -u
TAG=${1:-r01d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles
RAW=/tmp/e4s_prof_q
mkdir -p "$OUT" "$RAW"
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $RAW/probe -o probe -- python bench.py --probe-only --probe-reps 50 > $RAW/probe.log 2>&1; echo "probe rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $RAW/bench -o bench -- python bench.py --steps 4 --warmup 1 --steps-only --no-graph > $RAW/bench.log 2>&1; echo "bench rc=$?"
python tools/prof_summarize.py trace "$(find $RAW/probe -name '*_kernel_trace.csv' | head -1)" > $OUT/${TAG}_probe_kernel_stats.csv
python tools/prof_summarize.py trace "$(find $RAW/bench -name '*_kernel_trace.csv' | head -1)" > $OUT/${TAG}_bench_kernel_stats.csv
grep -h '^{' $RAW/probe.log $RAW/bench.log > $OUT/${TAG}_bench_lines.json
head -4 $OUT/${TAG}_probe_kernel_stats.csv
