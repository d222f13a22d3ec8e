"""Average duration per sort kernel from a rocprofv3 kernel trace (csv) of tools/bench_sort.py."""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "sort" in n or "tile_" in n:
        d[(re.search(r"::(\w+<[^>]*>)", n).group(1), r.get("Grid_Size_X") or r.get("Grid_Size"))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(k, len(v), round(sum(v) / len(v), 2))
