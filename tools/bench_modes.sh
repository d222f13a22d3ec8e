#!/bin/bash
# compare step time across graph / overlap settings
for g in 0 1; do for o in 0 1; do
  timeout -k 10 200 python -u bench.py --steps 50 --warmup 10 --cpu-baseline 0 --graph $g --overlap $o 2>/dev/null > gpurun_out/mode_${g}_${o}.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/mode_${g}_${o}.json").read().strip().splitlines()[-1])
print("graph", $g, "overlap", $o, round(d["value"]), round(d["ms_per_step"],4))
PY
done; done
