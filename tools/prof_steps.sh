#!/bin/bash
# usage: tools/prof_steps.sh <name> <marker> [bench.py arguments ...]
# Kernel trace of a bench.py run: the per-step breakdown of its fastest steps (the timed graph replays) and the timeline
# of one of the last steps, delimited by the kernel whose name contains <marker>.
set -u
NAME=$1; MARK=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$NAME
rm -rf "$OUT" && mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof" -o bench -- \
    python3 "$ROOT/bench.py" --cpu-baseline 0 --event-pass 0 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" ) || exit 1
trace=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python "$ROOT/tools/trace_steps.py" "$trace" ${KEEP:-15} "$MARK" > "$OUT/breakdown.txt" || exit 1
n=$(grep -c "$MARK" "$trace")
python "$ROOT/tools/trace_one_step.py" "$trace" "$MARK" $((n - 4)) 1 > "$OUT/one_step.txt" || exit 1
rm -rf "$OUT/prof"
cat "$OUT/breakdown.txt" "$OUT/one_step.txt"
