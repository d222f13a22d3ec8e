"""Launch durations of the AFM attention-pooling kernels at the Criteo_x1 shape (F=39, d=16), graph-replayed.

Run on the GPU box:  python tools/bench_afm.py [A] [B ...]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recommendsystem_amd import ops
from tools.bench_interact import graph_time

F, d = 39, 16
A = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batches = [int(x) for x in sys.argv[2:]] or [4096, 16384]
P = F * (F - 1) // 2
for B in batches:
    g = torch.Generator().manual_seed(0)
    emb = (torch.randn(B, F, d, generator=g) * 0.3).cuda()
    W, bA, h = (torch.randn(A, d, generator=g) * 0.3).cuda(), torch.zeros(A).cuda(), torch.randn(A, generator=g).cuda()
    pw, pb, do = torch.randn(d, generator=g).cuda(), torch.zeros(1).cuda(), torch.randn(B, generator=g).cuda()
    flops = B * P * (d + 2 * d * A + 2 * A + 2 * d)
    for name, fn, fl in (("afm_fwd", lambda: ops.afm_fwd(emb, W, bA, h, pw, pb), flops),
                         ("afm_bwd", lambda: ops.afm_bwd(emb, do, W, bA, h, pw), 3 * flops)):
        us = graph_time(fn)
        print(json.dumps({"kernel": name, "A": A, "batch": B, "avg_us": round(us, 2), "TFLOP_per_s": round(fl / us / 1e6, 2)}), flush=True)
