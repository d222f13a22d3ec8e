"""Per-step kernel breakdown from a rocprofv3 kernel trace of bench.py: steps are delimited by the fused gather
launch (alone or sharing its launch with the index sort) or, with a third argument, by the kernel whose name contains
it (one launch per step, e.g. field_sort_kernel for the generic trainer); the fastest `keep` steps (the graph replays
of the timed region) are averaged.

    python tools/trace_steps.py <kernel_trace.csv> [keep] [marker]
"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
keep = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marker = sys.argv[3] if len(sys.argv) > 3 else None
starts = [i for i, r in enumerate(rows)
          if (marker in r[2] if marker else ("gather_fused_kernel" in r[2] or "sort_gather_kernel" in r[2]))]
steps = []
for a, b in zip(starts, starts[1:]):
    steps.append((rows[b][0] - rows[a][0], a, b))
steps.sort()
sel = steps[:keep]
print(f"{len(steps)} steps in trace; fastest {len(sel)}: wall {sum(s[0] for s in sel) / len(sel) / 1e3:.1f} us/step "
      f"(min {sel[0][0] / 1e3:.1f}, max {sel[-1][0] / 1e3:.1f})")
agg = defaultdict(lambda: [0, 0.0])
busy = 0.0
for _, a, b in sel:
    for s, e, n in rows[a:b]:
        short = n.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0][:70]
        agg[short][0] += 1
        agg[short][1] += (e - s) / 1e3
        busy += (e - s) / 1e3
print(f"sum of kernel durations: {busy / len(sel):.1f} us/step")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / len(sel):8.1f} us/step  {c / len(sel):5.1f} launches  avg {t / c:7.2f} us  {n}")
