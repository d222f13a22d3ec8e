#!/bin/bash
# SQ counters of one GEMM shape (two passes of <= 8 counters).  usage: pmc_gemm.sh amc bnc M N K tile sk
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY"
i=0
for P in "$P1" "$P2"; do
    i=$((i+1))
    OUT=$ROOT/gpurun_out/pmc_gemm_$i
    rm -rf "$OUT" && mkdir -p "$OUT"
    timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d "$OUT" -o pmc -- python3 "$ROOT/tools/gemm_one.py" "$@" 6 > "$OUT/log.txt" 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for i in (1, 2):
    f = glob.glob("$ROOT/gpurun_out/pmc_gemm_%d/**/*counter_collection.csv" % i, recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_f32" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:28s} {sum(v[1:]) / max(len(v) - 1, 1):16.0f}  (n={len(v)})")
PY
