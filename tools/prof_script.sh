#!/bin/bash
# rocprofv3 kernel stats of a python script on the GPU box.
# usage: tools/prof_script.sh <name> <script.py> [arguments ...]   -> gpurun_out/prof_<name>/kernel_stats.csv
set -u
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$NAME
rm -rf "$OUT" && mkdir -p "$OUT"
SCRIPT=$ROOT/$1; shift
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o run -- \
    python3 "$SCRIPT" "$@" > "$OUT/out.txt" 2> "$OUT/err.txt" ) || { tail -20 "$OUT/err.txt"; exit 1; }
find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/prof"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    print(f'{float(r["AverageNs"]) / 1e3:9.2f} us  x{r["Calls"]:>5}  min {float(r["MinNs"]) / 1e3:8.2f}  {n}')
PY
