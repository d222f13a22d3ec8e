"""Per-step wall time, kernel time and the longest kernels over the whole run, from a rocprofv3 kernel trace
(steps delimited by the kernel whose name contains the marker).

    python tools/trace_timeline.py <kernel_trace.csv> <marker> [every]
"""
import csv
import sys
from collections import defaultdict

path, marker = sys.argv[1], sys.argv[2]
every = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
print(f"{len(starts)} steps")
for n, (a, b) in enumerate(zip(starts, starts[1:])):
    if n % every:
        continue
    wall = (rows[b][0] - rows[a][0]) / 1e3
    agg = defaultdict(float)
    busy = 0.0
    gap = 0.0
    prev_end = rows[a][0]
    for s, e, k in rows[a:b]:
        short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        agg[short] += (e - s) / 1e3
        busy += (e - s) / 1e3
        gap = max(gap, (s - prev_end) / 1e3)
        prev_end = max(prev_end, e)
    top = sorted(agg.items(), key=lambda kv: -kv[1])[:6]
    print(f"step {n:4d}: wall {wall:8.1f} us  kernels {busy:8.1f} us  longest gap {gap:7.1f} us  " +
          "  ".join(f"{k}={v:.0f}" for k, v in top))
