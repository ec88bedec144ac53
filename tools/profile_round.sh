#!/bin/bash
# Run on the GPU box through gpurun: rocprofv3 kernel-trace/stats of bench.py (eager launches AND the graph replay the driver times) and of
# the headline-kernel probe, plus separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) on the probe AND on the timed steps (every kernel of
# the step gets HBM counters: the 512^2 / 1024^2 generator tail, VERDICT r3 'next' 1).  Raw output stays in /tmp; only condensed summaries are
# written to gpurun_out/profiles/ (copy them to profiles/ and commit).   usage: tools/profile_round.sh r04 [quick]
set -u
TAG=${1:-r01}
MODE=${2:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles
RAW=/tmp/e4s_prof
mkdir -p "$OUT" "$RAW"
run() { name=$1; shift; timeout 400 rocprofv3 "$@" > "$RAW/$name.log" 2>&1; echo "$name rc=$?"; }
STEPS="python bench.py --steps 6 --warmup 2 --steps-only"
run probe --kernel-trace --stats -f csv -d $RAW/probe -o probe -- python bench.py --probe-only --probe-reps 50
run bench --kernel-trace --stats -f csv -d $RAW/bench -o bench -- $STEPS --no-graph
run graph --kernel-trace --stats -f csv -d $RAW/graph -o graph -- $STEPS
if [ "$MODE" != quick ]; then
run fetch --kernel-trace --pmc FETCH_SIZE -f csv -d $RAW/fetch -o fetch -- python bench.py --probe-only --probe-reps 10
run write --kernel-trace --pmc WRITE_SIZE -f csv -d $RAW/write -o write -- python bench.py --probe-only --probe-reps 10
run sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -f csv -d $RAW/sq -o sq -- python bench.py --probe-only --probe-reps 10
S2="python bench.py --steps 2 --warmup 1 --steps-only --no-graph"
run sfetch --kernel-trace --pmc FETCH_SIZE -f csv -d $RAW/sfetch -o sfetch -- $S2
run swrite --kernel-trace --pmc WRITE_SIZE -f csv -d $RAW/swrite -o swrite -- $S2
run ssq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -f csv -d $RAW/ssq -o ssq -- $S2
fi
f() { find $RAW/$1 -name "*$2" 2>/dev/null | head -1; }
python tools/prof_summarize.py trace "$(f probe _kernel_trace.csv)" > $OUT/${TAG}_probe_kernel_stats.csv
python tools/prof_summarize.py trace "$(f bench _kernel_trace.csv)" > $OUT/${TAG}_bench_kernel_stats.csv
python tools/prof_summarize.py replay "$(f graph _kernel_trace.csv)" $OUT/${TAG}_graph_replay_info.json > $OUT/${TAG}_graph_replay_kernel_stats.csv   # replays ONLY
cp "$(f probe _kernel_stats.csv)" $OUT/${TAG}_probe_rocprof_stats.csv 2>/dev/null
cp "$(f bench _kernel_stats.csv)" $OUT/${TAG}_bench_rocprof_stats.csv 2>/dev/null
if [ "$MODE" != quick ]; then
for c in fetch write sq; do python tools/prof_summarize.py pmc "$(f $c _counter_collection.csv)" > $OUT/${TAG}_pmc_$c.csv; done
for c in sfetch swrite ssq; do python tools/prof_summarize.py pmc "$(f $c _counter_collection.csv)" grid > $OUT/${TAG}_steps_pmc_${c#s}.csv; done
python tools/steps_summary.py $OUT/${TAG} > $OUT/${TAG}_steps_per_kernel_summary.json      # one row per kernel: bytes, TB/s, MFMA busy, LDS conflicts
fi
grep -h '^{' $RAW/probe.log $RAW/bench.log $RAW/graph.log > $OUT/${TAG}_bench_lines.json
ls -la $OUT; du -sh $RAW
