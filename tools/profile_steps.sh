#!/bin/bash
# rocprofv3 kernel trace of the timed bench steps only (B=8 swaps/step, eager launches) -> per-kernel summary
set -u
TAG=${1:-r01b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles
RAW=/tmp/e4s_prof_steps
mkdir -p "$OUT" "$RAW"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $RAW/bench -o bench -- python bench.py --steps 4 --warmup 1 --steps-only --no-graph ${BENCH_ARGS:-} > $RAW/bench.log 2>&1
echo "rc=$?"
python tools/prof_summarize.py trace "$(find $RAW/bench -name '*_kernel_trace.csv' | head -1)" > $OUT/${TAG}_steps_kernel_stats.csv
grep -h '^{' $RAW/bench.log > $OUT/${TAG}_steps_bench_line.json
head -50 $OUT/${TAG}_steps_kernel_stats.csv
