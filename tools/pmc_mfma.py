"""mfma_util.json from one SQ counter pass over bench.py (tools/refresh_profiles.sh): matrix-core utilisation of every
kernel that issues matrix instructions.

SQ_VALU_MFMA_BUSY_CYCLES adds up, over all SIMDs, the cycles a matrix instruction occupies the matrix pipeline
(32 per v_mfma_f32_16x16x4_f32); SQ_BUSY_CU_CYCLES adds up, over all CUs, the cycles a CU holds at least one wave.
Per kernel and launch:
  mfma_busy_per_simd_cycles = SQ_VALU_MFMA_BUSY_CYCLES / 1024            (256 CUs x 4 SIMDs)
  mfma_util_of_busy_cus     = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)
SQ_INSTS_MFMA (matrix instructions issued, per wave) is kept raw.
The wave-level counters (SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_VALU) count quad-cycles and are
given as shares of SQ_WAVE_CYCLES."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")


def main():
    files = glob.glob(os.path.join(BASE, "pmc_sq", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter csv under {BASE}/pmc_sq")
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            a = agg[name][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    kernels = {}
    for k, c in agg.items():
        mean = {n: v[1] / max(v[0], 1) for n, v in c.items()}
        if mean.get("SQ_INSTS_MFMA", 0.0) <= 0:
            continue
        wave = max(mean.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        kernels[k] = {
            "launches": max(v[0] for v in c.values()),
            "SQ_INSTS_MFMA": mean["SQ_INSTS_MFMA"],
            "SQ_VALU_MFMA_BUSY_CYCLES": mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
            "SQ_BUSY_CU_CYCLES": mean.get("SQ_BUSY_CU_CYCLES", 0.0),
            "mfma_busy_per_simd_cycles": mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0,
            "mfma_util_of_busy_cus": mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(4.0 * mean.get("SQ_BUSY_CU_CYCLES", 0.0), 1.0),
            "wave_share_waiting": mean.get("SQ_WAIT_ANY", 0.0) / wave,
            "wave_share_issue_stalled": mean.get("SQ_WAIT_INST_ANY", 0.0) / wave,
            "wave_share_valu_active": mean.get("SQ_ACTIVE_INST_VALU", 0.0) / wave,
        }
    out = {"source": "rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
                     "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU, one pass of bench.py --graph 0, no tracing",
           "kernels": kernels}
    path = os.path.join(BASE, "mfma_util.json")
    json.dump(out, open(path, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["SQ_INSTS_MFMA"]):
        print(f"{v['launches']:4d}  mfma util {v['mfma_util_of_busy_cus']:.3f}  busy/SIMD {v['mfma_busy_per_simd_cycles']:10.0f} cyc  "
              f"waiting {v['wave_share_waiting']:.2f}  issue-stalled {v['wave_share_issue_stalled']:.2f}  {k[:60]}")


if __name__ == "__main__":
    main()
