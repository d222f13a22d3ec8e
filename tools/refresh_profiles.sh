#!/bin/bash
# Everything under profiles/ in one GPU-box call: default bench line, its rocprofv3 kernel trace and step breakdown,
# the bench variants and the interaction-kernel rooflines.  Results land in gpurun_out/; copy what is to be kept.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
timeout -k 10 400 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err || exit 1
tools/profile_bench.sh > gpurun_out/profile_bench.log 2>&1 || exit 1
tools/bench_variants.sh 2> gpurun_out/bench_variants.err || exit 1
tools/profile_interact.sh > gpurun_out/profile_interact.log 2>&1 || exit 1
echo refreshed
