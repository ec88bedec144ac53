#!/bin/bash
# Run on the GPU box through gpurun: rocprofv3 kernel-trace/stats of bench.py and of the headline-kernel
# probe, plus separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) on the probe.  Raw output stays in /tmp;
# only condensed summaries are written to gpurun_out/profiles/ (copy them to profiles/ and commit).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles
RAW=/tmp/e4s_prof
mkdir -p "$OUT" "$RAW"
run() { name=$1; shift; timeout 300 rocprofv3 "$@" > "$RAW/$name.log" 2>&1; echo "$name rc=$?"; }
run probe --kernel-trace --stats -f csv -d $RAW/probe -o probe -- python bench.py --probe-only --probe-reps 50
run bench --kernel-trace --stats -f csv -d $RAW/bench -o bench -- python bench.py --steps 4 --warmup 1 --steps-only --no-graph
run fetch --kernel-trace --pmc FETCH_SIZE -f csv -d $RAW/fetch -o fetch -- python bench.py --probe-only --probe-reps 10
run write --kernel-trace --pmc WRITE_SIZE -f csv -d $RAW/write -o write -- python bench.py --probe-only --probe-reps 10
run sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -f csv -d $RAW/sq -o sq -- python bench.py --probe-only --probe-reps 10
f() { find $RAW/$1 -name "*$2" | head -1; }
python tools/prof_summarize.py trace "$(f probe _kernel_trace.csv)" > $OUT/${TAG}_probe_kernel_stats.csv
python tools/prof_summarize.py trace "$(f bench _kernel_trace.csv)" > $OUT/${TAG}_bench_kernel_stats.csv
cp "$(f probe _kernel_stats.csv)" $OUT/${TAG}_probe_rocprof_stats.csv 2>/dev/null
cp "$(f bench _kernel_stats.csv)" $OUT/${TAG}_bench_rocprof_stats.csv 2>/dev/null
for c in fetch write sq; do python tools/prof_summarize.py pmc "$(f $c _counter_collection.csv)" > $OUT/${TAG}_pmc_$c.csv; done
grep -h '^{' $RAW/probe.log $RAW/bench.log > $OUT/${TAG}_bench_lines.json
ls -la $OUT; du -sh $RAW
