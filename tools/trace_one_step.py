"""Every kernel of one step of a rocprofv3 kernel trace, with start offsets: where the step waits.

    python tools/trace_one_step.py <kernel_trace.csv> <marker> <step number> [steps]
"""
import csv
import sys

path, marker, which = sys.argv[1], sys.argv[2], int(sys.argv[3])
count = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "") + "/" + r.get("Queue_Id", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
a, b = starts[which], starts[which + count]
t0 = rows[a][0]
prev_end = t0
for s, e, k, q in rows[a:b]:
    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    print(f"+{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  q{q}  {short}")
    prev_end = max(prev_end, e)
print(f"step wall {(rows[b][0] - t0) / 1e3:.1f} us")
