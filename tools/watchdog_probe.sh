#!/bin/bash
# Runs the one-rank RCCL test worker (tests/dist_worker.py gpu1) up to N times and stops at the first run that does not
# end with status 0, leaving that run's output in gpurun_out/probe/fail.log: where the process was when torch's
# process-group watchdog aborted it.      usage: tools/watchdog_probe.sh [N] [kind] [optimizer]
N=${1:-8}; KIND=${2:-pnn}; OPT=${3:-SGD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/probe
mkdir -p "$OUT"
for i in $(seq 1 "$N"); do
    d=$(mktemp -d)
    echo "run $i" >> "$OUT/progress.log"
    RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + i)) PYTHONPATH=$ROOT:$ROOT/tests \
        timeout -k 10 200 python -u "$ROOT/tests/dist_worker.py" gpu1 "$d" "$KIND" "$OPT" > "$OUT/run.log" 2>&1
    rc=$?
    rm -rf "$d"
    if [ $rc -ne 0 ]; then cp "$OUT/run.log" "$OUT/fail.log"; echo "run $i failed with status $rc" >> "$OUT/progress.log"; exit 0; fi
done
echo "no failure in $N runs" >> "$OUT/progress.log"
