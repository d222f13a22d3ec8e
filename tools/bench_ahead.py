"""Stand-alone timings of the pieces of the "next batch inside this step" path at the benchmark shape (39 fields,
B = 4096): the 1024-thread sort launch, the 256-thread sort alone in a grouped launch, the layer-1 backward products
with and without the sort riding along, and the per-distinct-row Adam catch-up.  Run under rocprofv3 --kernel-trace
--stats (tools/prof_script.sh) and read the kernel averages.

    python tools/bench_ahead.py [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recommendsystem_amd import data, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
vocab = data.criteo_x1_vocab_sizes()
B, F, d = 4096, len(vocab), 16
layout = ops.FieldLayout(vocab, dev)
idxs, _ = data.synthetic_batches(vocab, B, 4, "cpu", seed=1)
idxs = [t.to(dev) for t in idxs]
plan = ops.SparsePlan(layout, cap=B, d=d, with_lr=True)
plan2 = ops.SparsePlan(layout, cap=B, d=d, with_lr=True)
dz = torch.randn(B, 1024, device=dev)
x = torch.randn(B, 624, device=dev).relu_()
W = torch.randn(1024, 624, device=dev)
gW = torch.empty(1024, 624, device=dev)
ws = ops.Workspace(dev)
R = layout.total_rows
w = torch.randn(R, d, device=dev); m = torch.zeros_like(w); v = torch.zeros_like(w)
w1 = torch.randn(R, device=dev); m1 = torch.zeros_like(w1); v1 = torch.zeros_like(w1)
hyper = ops.make_hyper(dev, 1e-3)
step = torch.full((1,), 30, dtype=torch.int64, device=dev)
rs = ops.RowAdam(w, m, v, "Adam", hyper, step)
rs1 = ops.RowAdam(w1, m1, v1, "Adam", hyper, step, bctab=rs.bctab)


def products(sort=None):
    skp = ops.SplitKParts(ws)
    grp = ops.GemmGroup(3)
    grp.add(dz, x, a_mc=True, b_nc=True, out=gW, defer_reduce=skp)
    grp.add(dz, W, b_nc=True)
    grp.launch(sort=sort)
    skp.finish()


dz2 = torch.randn(B, 512, device=dev)
x2 = torch.randn(B, 1024, device=dev).relu_()
W2 = torch.randn(512, 1024, device=dev)
gW2 = torch.empty(512, 1024, device=dev)


def products2(sort=None):
    skp = ops.SplitKParts(ws, tag="b")
    grp = ops.GemmGroup(3)
    grp.add(dz2, x2, a_mc=True, b_nc=True, out=gW2, defer_reduce=skp)
    grp.add(dz2, W2, b_nc=True)
    grp.launch(sort=sort)
    skp.finish()


def section(name, fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(0)
    torch.cuda.synchronize()
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / reps:.1f} us per iteration (eager, back to back)", flush=True)


section("1024-thread sort launch", lambda i: plan.sort_columnar(idxs[i % 4]))
section("256-thread sort alone (grouped launch without products)", lambda i: plan2._sort_in_group(None, 0, idxs[i % 4]))
section("layer-1 products", lambda i: products())
section("layer-1 products + whole sort", lambda i: products(sort=(plan2, idxs[i % 4], 0)))
section("layer-2 products", lambda i: products2())
section("layer-2 products + sort phase 1", lambda i: products2(sort=(plan2, idxs[i % 4], 1)))
section("layer-1 products + sort phase 2", lambda i: products(sort=(plan2, idxs[i % 4], 2)))


def cplan(i):
    rs.last.zero_(); rs1.last.zero_()
    plan.catchup_plan(rs, rs1, mode=0)


def centry(i):
    rs.last.zero_(); rs1.last.zero_()
    ops.adam_catchup(idxs[0], layout, rs, rs1, 0)


plan.sort_columnar(idxs[0])
section("catch-up per distinct row (+2 fills)", cplan)
section("catch-up per entry (+2 fills)", centry)
print("done")
