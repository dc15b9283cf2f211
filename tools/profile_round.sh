#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + separate PMC passes of the bench command,
# summaries written to gpurun_out/profile_<tag>/.  Counters are collected in their own runs
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass) with --kernel-trace only.
# "headline": the timed W4A16 8192^2 step only (--no-extra --one-stream: kernels do not overlap, so the
# per-kernel durations are comparable with the HIP-event figures of bench.py), not mixed
# with the small TinyLlama-shaped launches; "extras": the whole default bench (all legs).
#   usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
HEAD="python $ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extra --one-stream"
FULL="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
run() {  # name, rocprof args..., -- cmd
    local name=$1; shift
    rocprofv3 --kernel-trace "$@" > "$OUT/$name.stdout" 2> /dev/null
    { echo "# srchash $(cat "$ROOT/compressed_tensors_amd/libct_hip.so.srchash" 2>/dev/null)"; echo "# cmd rocprofv3 --kernel-trace $*";
      python "$ROOT/tools/prof_summary.py" /tmp/prof_$TAG/$name/run_results.db; } > "$OUT/$name.txt"
}
cp "$ROOT/compressed_tensors_amd/libct_hip.so.srchash" "$OUT/srchash" 2>/dev/null
run headline_trace --stats -d /tmp/prof_$TAG/headline_trace -o run -- $HEAD
export CT_BENCH_WARM_SCALE=0.1  # counter passes: per-kernel counters do not depend on clocks
run headline_fetch --pmc FETCH_SIZE -d /tmp/prof_$TAG/headline_fetch -o run -- $HEAD
run headline_write --pmc WRITE_SIZE -d /tmp/prof_$TAG/headline_write -o run -- $HEAD
run headline_sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/prof_$TAG/headline_sq -o run -- $HEAD
CT_BENCH_WARM_SCALE=1 run extras_trace --stats -d /tmp/prof_$TAG/extras_trace -o run -- $FULL
run extras_fetch --pmc FETCH_SIZE -d /tmp/prof_$TAG/extras_fetch -o run -- $FULL
run extras_write --pmc WRITE_SIZE -d /tmp/prof_$TAG/extras_write -o run -- $FULL
rm -f "$OUT"/*_fetch.stdout "$OUT"/*_write.stdout "$OUT"/*_sq.stdout
ls -la "$OUT"
