"""pmc_traffic.json (written beside the passes; tools/collect_profiles.py copies it into profiles/) from the two counter
passes of tools/refresh_profiles.sh.

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE counts 128-byte read requests as
64 bytes, so it is doubled.  Per kernel name: launches, mean read / written bytes per launch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


BASE = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")


def load(d, counter):
    files = glob.glob(os.path.join(BASE, d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter csv under {BASE}/{d}")
    agg = defaultdict(lambda: [0, 0.0])
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            a = agg[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"]) * 1024.0
    return agg


def main():
    rd, wr = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(rd) | set(wr)):
        n = max(rd[k][0], wr[k][0])
        kernels[k] = {"launches": n, "read_bytes_per_launch": 2.0 * rd[k][1] / max(rd[k][0], 1),
                      "written_bytes_per_launch": wr[k][1] / max(wr[k][0], 1)}
    gemm = [v for k, v in kernels.items() if k.startswith("gemm_f32")]
    tot_l = sum(v["launches"] for v in gemm)
    dom = sum(v["launches"] * (v["read_bytes_per_launch"] + v["written_bytes_per_launch"]) for v in gemm) / max(tot_l, 1)
    out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of bench.py --graph 0 --share-launches 0 (tools/refresh_profiles.sh)",
           "corrections": "values are KiB per dispatch; FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B)",
           "dominant_kernel": "gemm_f32_dma_kernel (all MLP GEMM launches of a step)",
           "dominant_kernel_hbm_bytes_per_launch": dom, "kernels": kernels}
    path = os.path.join(BASE, "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, "dominant bytes/launch:", dom)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["read_bytes_per_launch"])[:14]:
        print(f"{v['launches']:5d}  rd {v['read_bytes_per_launch'] / 1e6:9.2f} MB  wr {v['written_bytes_per_launch'] / 1e6:9.2f} MB  {k[:70]}")


if __name__ == "__main__":
    main()
