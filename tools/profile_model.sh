#!/bin/bash
# Kernel-time breakdown of one bench.py variant (generic touched-rows trainer): rocprofv3 kernel trace + per-kernel stats.
# usage: tools/profile_model.sh <tag> <bench.py arguments...>     -> gpurun_out/model_prof/<tag>_stats.csv, <tag>.json
set -u
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/model_prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$TAG" -o p -- \
    python3 "$ROOT/bench.py" "$@" --cpu-baseline 0 --event-pass 0 > "$OUT/$TAG.json" 2> "$OUT/$TAG.err" || exit 1
find "$OUT/$TAG" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_stats.csv" \;
python3 "$ROOT/tools/trace_steps.py" "$(find "$OUT/$TAG" -name "*kernel_trace.csv" | head -1)" 15 sort > "$OUT/${TAG}_steps.txt"
rm -rf "$OUT/$TAG"
