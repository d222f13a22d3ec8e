"""Experiment: duration of the riding sort inside the layer-1 backward launch (library built with -DRS_SORT_TIMING)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommendsystem_amd import data, ops
dev = torch.device("cuda", 0)
vocab = data.criteo_x1_vocab_sizes()
B, F, d = 4096, len(vocab), 16
layout = ops.FieldLayout(vocab, dev)
idxs, _ = data.synthetic_batches(vocab, B, 4, "cpu", seed=1)
idxs = [t.to(dev) for t in idxs]
plan2 = ops.SparsePlan(layout, cap=B, d=d, with_lr=True)
dz = torch.randn(B, 1024, device=dev); x = torch.randn(B, 624, device=dev).relu_(); W = torch.randn(1024, 624, device=dev)
gW = torch.empty(1024, 624, device=dev); ws = ops.Workspace(dev)
def products(sort=None):
    skp = ops.SplitKParts(ws); grp = ops.GemmGroup(3)
    grp.add(dz, x, a_mc=True, b_nc=True, out=gW, defer_reduce=skp); grp.add(dz, W, b_nc=True)
    grp.launch(sort=sort); skp.finish()
dz2 = torch.randn(B, 512, device=dev); x2 = torch.randn(B, 1024, device=dev).relu_(); W2 = torch.randn(512, 1024, device=dev)
gW2 = torch.empty(512, 1024, device=dev)
def products2(sort=None):
    skp = ops.SplitKParts(ws, tag="b"); grp = ops.GemmGroup(3)
    grp.add(dz2, x2, a_mc=True, b_nc=True, out=gW2, defer_reduce=skp); grp.add(dz2, W2, b_nc=True)
    grp.launch(sort=sort); skp.finish()
off = F * (6 * B + 2 + 4) + 2 * F * B           # ints: behind the plan arrays and the spilled items [F, cap] x 64 bit
def show(tag, phase2):
    t = plan2.plan[off + (4 * F if phase2 else 0): off + (4 * F if phase2 else 0) + 4 * F].view(torch.int64)[:2 * F].view(F, 2).cpu()
    dur = (t[:, 1] - t[:, 0]).float() / 100.0
    print(tag, "sort us per field: max %.1f mean %.1f; every 6th %s" % (dur.max(), dur.mean(), [round(float(x), 1) for x in dur[:39:6]]))
for i in range(3):
    plan2._sort_in_group(None, 0, idxs[i % 4], 1); torch.cuda.synchronize(); show("phase 1 alone", False)
    plan2._sort_in_group(None, 0, idxs[i % 4], 2); torch.cuda.synchronize(); show("phase 2 alone", True)
for i in range(3):
    products2(sort=(plan2, idxs[i % 4], 1)); torch.cuda.synchronize(); show("phase 1 beside the layer-2 products", False)
    products(sort=(plan2, idxs[i % 4], 2)); torch.cuda.synchronize(); show("phase 2 beside the layer-1 products", True)
