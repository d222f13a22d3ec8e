#!/bin/bash
# Bench lines beside the headline: cache-hostile uniform indices, the 26-field variant, a one-thread CPU baseline, and
# the interaction models through the generic touched-rows trainer.  One JSON line each -> gpurun_out/bench_variants.jsonl
set -u
out=gpurun_out/bench_variants.jsonl
: > $out
run() { echo "== $*" >&2; timeout -k 10 400 python bench.py "$@" >> $out || exit 1; }
run --uniform 1 --cpu-baseline 0
run --fields 26 --cpu-baseline 0
run --cpu-threads 1 --cpu-seconds 15 --event-pass 0
run --model fm --cpu-baseline 0
for m in ffm pnn_inner dcn nfm afm; do
  run --model $m --steps 20 --warmup 5 --cpu-baseline 0
done
# the batch size the reference's PNN / DCN configuration names (tiled index sort above 4096 entries per field)
for m in dcn pnn_inner; do
  run --model $m --batch 16384 --steps 10 --warmup 3 --cpu-baseline 0
done
run --batch 16384 --steps 20 --warmup 5 --cpu-baseline 0
run --model pnn --steps 5 --warmup 2 --cpu-baseline 0 --distinct-batches 2
