#!/bin/bash
# usage: tools/prof_one_step.sh <name> <marker> <step> [bench.py arguments ...]
set -u
NAME=$1; MARK=$2; STEP=$3; shift; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$NAME
rm -rf "$OUT" && mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof" -o bench -- \
    python3 "$ROOT/bench.py" --cpu-baseline 0 --event-pass 0 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" ) || exit 1
trace=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python "$ROOT/tools/trace_one_step.py" "$trace" "$MARK" "$STEP" ${COUNT:-1} > "$OUT/one_step.txt" || exit 1
rm -rf "$OUT/prof"
cat "$OUT/one_step.txt"
