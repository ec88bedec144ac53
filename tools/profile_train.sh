#!/bin/bash
# Run on the GPU box through gpurun: rocprofv3 kernel trace of the config-5 train legs (bench.py --train-only), condensed per kernel into
# gpurun_out/profiles/<tag>_train_kernel_stats.csv.
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles
RAW=/tmp/e4s_prof_train
mkdir -p "$OUT" "$RAW"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $RAW/train -o train -- python bench.py --train-only --train-steps 6 > $RAW/train.log 2>&1 < /dev/null
echo "rc=$?"
t=$(find $RAW/train -name "*_kernel_trace.csv" | head -1)
if [ -n "$t" ]; then python tools/prof_summarize.py trace "$t" > $OUT/${TAG}_train_kernel_stats.csv; head -45 $OUT/${TAG}_train_kernel_stats.csv | cut -c1-150; fi
grep -h '^{' $RAW/train.log > $OUT/${TAG}_train_line.json
