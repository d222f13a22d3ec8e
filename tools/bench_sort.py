"""Launch duration of the index sort alone (rs_sparse_bwd_sort of a columnar batch) at the bench's 39 vocabularies.

Run on the GPU box:  python tools/bench_sort.py [B ...]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recommendsystem_amd import ops
from tools.bench_interact import graph_time

vocabs = [100] * 13 + [int(round(10 ** (1 + 5 * i / 25))) for i in range(26)]
for B in [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096, 8192]:
    g = torch.Generator().manual_seed(0)
    idx = torch.stack([torch.randint(0, v, (B,), generator=g) for v in vocabs]).cuda()
    layout = ops.FieldLayout(vocabs, "cuda:0")
    plan = ops.SparsePlan(layout, cap=B, d=16, with_lr=True)
    us = graph_time(lambda: plan.sort_columnar(idx))
    print(json.dumps({"kernel": "index sort", "batch": B, "fields": len(vocabs), "avg_us": round(us, 2)}), flush=True)
