#!/bin/bash
# HBM traffic counters of the bench step, one counter per rocprofv3 pass (no tracing in these runs).
# Writes gpurun_out/pmc_fetch/ and gpurun_out/pmc_write/.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --graph 0 --cpu-baseline 0 --distinct-batches 2 --event-pass 0"
for c in FETCH_SIZE WRITE_SIZE; do
    OUT=$ROOT/gpurun_out/pmc_$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
    rm -rf "$OUT" && mkdir -p "$OUT"
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT" -o pmc -- python3 "$ROOT/bench.py" $ARGS \
        > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
done
ls $ROOT/gpurun_out/pmc_fetch $ROOT/gpurun_out/pmc_write
