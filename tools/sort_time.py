"""Time rs_sparse_bwd_sort alone for different key widths (how much of it is the radix passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommendsystem_amd import ops

def run(vocab, B=4096, F=39, reps=200):
    dev = "cuda:0"
    layout = ops.FieldLayout([vocab] * F, dev)
    plan = ops.SparsePlan(layout, cap=B, d=16, with_lr=True)
    idx = torch.randint(1, vocab, (F, B), device=dev)
    for _ in range(10):
        plan.sort_columnar(idx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.sort_columnar(idx)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for v in (4, 16, 256, 4096, 65536, 1 << 20):
    print(f"vocab {v:8d}: {run(v):7.2f} us (F=39)   {run(v, F=1):7.2f} us (F=1)")
