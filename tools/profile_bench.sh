#!/bin/bash
# Kernel-trace profile of the default bench (graph replay) on the GPU box; writes gpurun_out/prof/*.
# usage: tools/profile_bench.sh [extra bench args]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- \
    python3 "$ROOT/bench.py" --steps 50 --warmup 10 --cpu-baseline 0 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
ls "$OUT"
