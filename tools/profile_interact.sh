#!/bin/bash
# Per-kernel roofline lines of the interaction kernels plus a rocprofv3 kernel trace of the same command, on the GPU box.
# Writes gpurun_out/interact/{roofline.jsonl, interact_kernel_stats.csv}.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/interact
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 400 python3 "$ROOT/tools/bench_interact.py" 4096 16384 > "$OUT/roofline.jsonl" 2> "$OUT/bench.err"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o interact -- \
    python3 "$ROOT/tools/bench_interact.py" 4096 > "$OUT/roofline_under_rocprof.jsonl" 2> "$OUT/prof.err"
find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/interact_kernel_stats.csv" \;
ls "$OUT"
