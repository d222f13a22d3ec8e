#!/bin/bash
# exact vs lazy Adam on the default bench line, with 8 and 64 distinct resident batches, plus a kernel trace
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 200 python bench.py --steps 40 --warmup 10 --cpu-baseline 0 > gpurun_out/r02/bench_exact.json 2> gpurun_out/r02/bench_exact.err || exit 1
timeout -k 10 200 python bench.py --steps 40 --warmup 10 --cpu-baseline 0 --exact-adam 0 > gpurun_out/r02/bench_lazy.json 2> gpurun_out/r02/bench_lazy.err || exit 1
timeout -k 10 300 python bench.py --steps 128 --warmup 64 --cpu-baseline 0 --distinct-batches 64 --event-pass 0 > gpurun_out/r02/bench_exact_nb64.json 2> gpurun_out/r02/bench_exact_nb64.err || exit 1
tools/profile_bench.sh --distinct-batches 64 --steps 128 --warmup 64 --event-pass 0 > gpurun_out/r02/profile.log 2>&1 || exit 1
echo done
