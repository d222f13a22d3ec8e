#!/bin/bash
# Kernel trace of one bench.py configuration on the GPU box: per-kernel stats + per-step breakdown.
# usage: tools/prof_bench.sh <name> [bench.py arguments ...]   -> gpurun_out/prof_<name>/{bench.json,kernel_stats.csv,steps.txt}
set -u
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$NAME
rm -rf "$OUT" && mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- \
    python3 "$ROOT/bench.py" --cpu-baseline 0 --event-pass 0 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" ) || exit 1
find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
trace=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python "$ROOT/tools/trace_steps.py" "$trace" ${KEEP:-20} ${MARKER:-} > "$OUT/steps.txt" || exit 1
rm -rf "$OUT/prof"
cat "$OUT/steps.txt"
