#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + separate PMC passes of the default bench
# command, summaries written to gpurun_out/profile_<tag>/.  Counters are collected in their own
# runs (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass) with --kernel-trace only.
#   usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/trace -o run -- $CMD > "$OUT/bench_under_trace.json" 2> /dev/null
python "$ROOT/tools/prof_summary.py" /tmp/prof_$TAG/trace/run_results.db > "$OUT/kernel_trace_stats.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_$TAG/fetch -o run -- $CMD > /dev/null 2>&1
python "$ROOT/tools/prof_summary.py" /tmp/prof_$TAG/fetch/run_results.db > "$OUT/pmc_fetch.txt"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_$TAG/write -o run -- $CMD > /dev/null 2>&1
python "$ROOT/tools/prof_summary.py" /tmp/prof_$TAG/write/run_results.db > "$OUT/pmc_write.txt"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/prof_$TAG/sq -o run -- $CMD > /dev/null 2>&1
python "$ROOT/tools/prof_summary.py" /tmp/prof_$TAG/sq/run_results.db > "$OUT/pmc_sq.txt"
ls -la "$OUT"
