#!/bin/bash
# usage: tools/prof_timeline.sh <name> <marker> [bench.py arguments ...]
set -u
NAME=$1; MARK=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$NAME
rm -rf "$OUT" && mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof" -o bench -- \
    python3 "$ROOT/bench.py" --cpu-baseline 0 --event-pass 0 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" ) || exit 1
trace=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python "$ROOT/tools/trace_timeline.py" "$trace" "$MARK" ${EVERY:-5} > "$OUT/timeline.txt" || exit 1
rm -rf "$OUT/prof"
cat "$OUT/timeline.txt"
