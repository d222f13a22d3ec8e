#!/bin/bash
# Everything that ends up under profiles/ for one round, in one GPU-box call.  Results land in gpurun_out/profiles_rNN/
# (the only directory a GPU call brings back); tools/collect_profiles.py then copies them into profiles/rNN_* in the
# repository and checks that every file is what it claims to be.
# usage: tools/refresh_profiles.sh r03 [sections]     sections: any of "main variants dist interact pmc" (default: all;
# a GPU call is limited to 20 minutes, so the sections usually go in two or three calls -- their files accumulate)
set -u
R=${1:-r03}
SECTIONS=${2:-"main variants dist interact pmc"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/profiles_$R
cd "$ROOT"
mkdir -p "$OUT"
step() { echo "== $*" >&2; }
want() { case " $SECTIONS " in *" $1 "*) return 0;; *) return 1;; esac; }

if want main; then
step "default bench line"
timeout -k 10 400 python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err" || exit 1

step "kernel trace of the same command"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- \
    python3 "$ROOT/bench.py" --cpu-baseline 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" ) || exit 1
find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
trace=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python tools/trace_steps.py "$trace" 50 > "$OUT/step_breakdown.txt" || exit 1
rm -rf "$OUT/prof"
fi

if want variants; then
step "bench variants"
V=$OUT/bench_variants.jsonl
: > "$V"
run() { echo "== bench.py $*" >&2; timeout -k 10 400 python bench.py "$@" >> "$V" 2>> "$OUT/bench_variants.err" || exit 1; }
run --exact-adam 0 --cpu-baseline 0                      # lazy Adam: what the exact replay costs
run --mode dense --cpu-baseline 0 --steps 20             # the reference's dense optimizer sweep
run --distinct-batches 64 --steps 128 --warmup 64 --cpu-baseline 0 --event-pass 0    # rows come back after 64 steps
run --steps 300 --warmup 10 --cpu-baseline 0 --event-pass 0    # a longer run: returning rows have more skipped steps to replay
run --age-steps 3000 --cpu-baseline 0 --event-pass 0           # the level a long run settles at: tables with 3000 steps of history
run --uniform 1 --cpu-baseline 0
run --fields 26 --cpu-baseline 0
run --cpu-threads 1 --cpu-seconds 15 --event-pass 0
run --model fm --cpu-baseline 0
for m in ffm pnn_inner dcn nfm afm; do
  run --model $m --steps 20 --warmup 5 --cpu-baseline 0
done
run --model pnn_inner --activation tanh --steps 20 --warmup 5 --cpu-baseline 0       # a 1-wide head that is not a dead ReLU
# FFM's replay of skipped Adam steps works on rows of 2.5 KB: how the step depends on how stale the rows are
run --model ffm --steps 100 --warmup 5 --cpu-baseline 0 --event-pass 0
run --model ffm --distinct-batches 64 --steps 128 --warmup 64 --cpu-baseline 0 --event-pass 0
run --model ffm --steps 20 --warmup 5 --exact-adam 0 --cpu-baseline 0 --event-pass 0
run --model ffm --steps 20 --warmup 5 --age-steps 1000 --cpu-baseline 0 --event-pass 0
run --optimizer SGD --regular 1e-4 --cpu-baseline 0 --event-pass 0                    # SGD + emb_regular on touched rows
run --optimizer SGD --regular 1e-4 --mode dense --steps 20 --cpu-baseline 0 --event-pass 0
for m in dcn pnn_inner; do
  run --model $m --batch 16384 --steps 10 --warmup 3 --cpu-baseline 0
done
run --batch 16384 --steps 20 --warmup 5 --cpu-baseline 0
run --model pnn --activation tanh --steps 5 --warmup 2 --cpu-baseline 0 --distinct-batches 2
run --model pnn --activation tanh --batch 16384 --steps 3 --warmup 1 --cpu-baseline 0 --distinct-batches 2 --event-pass 0
run --dropout 0.9 --cpu-baseline 0                       # the reference script's rate on the first block (DeepFM.py:61): mask in the GEMM epilogues

fi

if want dist; then
step "row-sharded step on one rank: every collective through RCCL (--force-dist 1)"
timeout -k 10 300 python bench.py --force-dist 1 --cpu-baseline 0 > "$OUT/force_dist_bench.json" 2> "$OUT/force_dist_bench.err" || exit 1
FV=$OUT/force_dist_variants.jsonl
: > "$FV"
frun() { echo "== bench.py --force-dist 1 $*" >&2; timeout -k 10 300 python bench.py --force-dist 1 --cpu-baseline 0 --event-pass 0 "$@" >> "$FV" 2>> "$OUT/force_dist_variants.err" || exit 1; }
frun                                     # the default again, without the event pass
frun --ahead 0                           # the next batch is not announced: sort and key exchange at the head of the step
frun --overlap-comm 0                    # everything on one stream
frun --dist-graph 0                      # eager launches
frun --tables replicated                 # replicated tables, all-gather of (row, gradient) pairs
frun --exact-adam 0                      # lazy Adam
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fprof" -o bench -- \
    python3 "$ROOT/bench.py" --force-dist 1 --cpu-baseline 0 --event-pass 0 > "$OUT/force_dist_under_rocprof.json" 2> "$OUT/force_dist_under_rocprof.err" ) || exit 1
find "$OUT/fprof" -name "*kernel_stats.csv" -exec cp {} "$OUT/force_dist_kernel_stats.csv" \;
trace=$(find "$OUT/fprof" -name "*kernel_trace.csv" | head -1)
python tools/trace_steps.py "$trace" 50 shard_lookup_kernel > "$OUT/force_dist_step_breakdown.txt" || exit 1
python tools/trace_one_step.py "$trace" shard_lookup_kernel 40 2 > "$OUT/force_dist_two_steps.txt" || exit 1
rm -rf "$OUT/fprof"
fi

if want interact; then
step "interaction kernels"
timeout -k 10 400 python tools/bench_interact.py 4096 16384 > "$OUT/interact_roofline.jsonl" 2> "$OUT/interact.err" || exit 1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/iprof" -o interact -- \
    python3 "$ROOT/tools/bench_interact.py" 4096 > /dev/null 2> "$OUT/interact_prof.err" ) || exit 1
find "$OUT/iprof" -name "*kernel_stats.csv" -exec cp {} "$OUT/interact_kernel_stats.csv" \;
rm -rf "$OUT/iprof"
fi

if want pmc; then
step "HBM traffic counters (one counter per pass, no tracing)"
ARGS="--steps 4 --warmup 2 --graph 0 --cpu-baseline 0 --event-pass 0 --share-launches 0"
for c in FETCH_SIZE WRITE_SIZE; do
    D=$OUT/pmc_$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
    mkdir -p "$D"
    ( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$D" -o pmc -- \
        python3 "$ROOT/bench.py" $ARGS > "$D/bench.json" 2> "$D/bench.err" ) || exit 1
done
python tools/pmc_traffic.py "$OUT" || exit 1
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write"

step "matrix-core utilisation (one SQ counter pass, no tracing)"
D=$OUT/pmc_sq
mkdir -p "$D"
( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
    --output-format csv -d "$D" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$D/bench.json" 2> "$D/bench.err" ) || exit 1
python tools/pmc_mfma.py "$OUT" || exit 1
rm -rf "$OUT/pmc_sq"
fi
ls -la "$OUT"
echo refreshed
