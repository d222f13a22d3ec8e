"""Roofline of each feature-interaction kernel at the Criteo_x1 shape (F=39, d=16), one JSON line per kernel.

Run on the GPU box:  python tools/bench_interact.py [B ...]   (default batches 4096 and 16384)

`achieved` = algorithmic bytes (HBM-bound kernels) or FLOPs (PNN products) of one launch / average launch
duration.  The duration is measured with HIP events over a HIP-graph replay of `iters` back-to-back launches, so
host launch latency is outside the measurement and the kernels run as they do inside a captured training step.
Algorithmic bytes count every input read once and every output written once:

    pnn_inner fwd   B*(D + P)*4                 bwd  B*(D + P + D)*4
    pnn_outer fwd   B*(D + P*d*d)*4             bwd  B*(D + P*d*d + D)*4
    dcn_cross fwd   B*(D + D)*4  (+ S)          bwd  B*(D + D + D)*4
    nfm_bi    fwd   B*(D + k)*4                 bwd  B*(D + k + D)*4
    afm       fwd   B*(D + 1)*4                 bwd  B*(D + 1 + D)*4
    ffm       fwd   B*F*8 + U*F*d*4 + B*4       bwd  fwd bytes + U*F*d*4 written     (U = distinct rows of the batch)

FFM: a sample reads F rows of F*d floats, but a row that several samples name comes from HBM once; the bytes the
workgroups pull through the caches (B*F*F*d*4) are reported beside it as `cache_side_MB` / `cache_side_GB_per_s`.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recommendsystem_amd import ops

HBM_PEAK = 8000.0       # GB/s
F, d = 39, 16
D, P = F * d, F * (F - 1) // 2


def graph_time(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(5):
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best


def report(name, B, us, byts, flops=None, note="", cache_bytes=None):
    line = {"kernel": name, "batch": B, "avg_us": round(us, 2), "algorithmic_MB": round(byts / 1e6, 2),
            "GB_per_s": round(byts / us / 1e3, 1), "frac_of_hbm_peak": round(byts / us / 1e3 / HBM_PEAK, 3)}
    if cache_bytes:
        line["cache_side_MB"] = round(cache_bytes / 1e6, 2)
        line["cache_side_GB_per_s"] = round(cache_bytes / us / 1e3, 1)
    if flops:
        line["GFLOP"] = round(flops / 1e9, 3)
        line["TFLOP_per_s"] = round(flops / us / 1e6, 2)
    if note:
        line["note"] = note
    print(json.dumps(line), flush=True)


def run(B):
    dev = "cuda:0"
    g = torch.Generator().manual_seed(2021)
    emb = (torch.randn(B, F, d, generator=g) * 0.1).to(dev)

    out = ops.pnn_inner_fwd(emb)
    dout = torch.randn_like(out)
    report("pnn_inner_fwd", B, graph_time(lambda: ops.pnn_inner_fwd(emb)), B * (D + P) * 4, flops=2.0 * B * P * d)
    report("pnn_inner_bwd", B, graph_time(lambda: ops.pnn_inner_bwd(emb, dout)), B * (2 * D + P) * 4, flops=4.0 * B * P * d)

    Bo = min(B, 4096)          # the materialised outer product is 759 KB per sample
    embo = emb[:Bo].contiguous()
    out = ops.pnn_outer_fwd(embo)
    dout = torch.randn_like(out)
    report("pnn_outer_fwd", Bo, graph_time(lambda: ops.pnn_outer_fwd(embo), 5), Bo * (D + P * d * d) * 4)
    report("pnn_outer_bwd", Bo, graph_time(lambda: ops.pnn_outer_bwd(embo, dout), 5), Bo * (2 * D + P * d * d) * 4)
    del out, dout

    x0 = emb.view(B, D)
    w = (torch.randn(3, D, generator=g) * 0.05).to(dev)
    b = (torch.randn(3, D, generator=g) * 0.05).to(dev)
    xo, S = ops.dcn_cross_fwd(x0, w, b)
    dxo = torch.randn_like(xo)
    report("dcn_cross_fwd(L=3)", B, graph_time(lambda: ops.dcn_cross_fwd(x0, w, b)), B * 2 * D * 4 + S.numel() * 4)
    report("dcn_cross_bwd(L=3)", B, graph_time(lambda: ops.dcn_cross_bwd(x0, S, dxo, w, b)), B * 3 * D * 4 + S.numel() * 4)

    V = torch.randn(D, 4, generator=g).to(dev)
    o, s = ops.nfm_bi_fwd(x0, V)
    do = torch.randn_like(o)
    report("nfm_bi_fwd(k=4)", B, graph_time(lambda: ops.nfm_bi_fwd(x0, V)), B * (D + 4) * 4)
    report("nfm_bi_bwd(k=4)", B, graph_time(lambda: ops.nfm_bi_bwd(x0, V, s, do)), B * (2 * D + 8) * 4)

    A = 8
    W = (torch.randn(A, d, generator=g) * 0.3).to(dev)
    bA = torch.zeros(A, device=dev)
    h = (torch.randn(1, A, generator=g) * 0.3).to(dev)
    pw = (torch.randn(1, d, generator=g) * 0.3).to(dev)
    pb = torch.zeros(1, device=dev)
    o = ops.afm_fwd(emb, W, bA, h, pw, pb)
    do = torch.randn_like(o)
    afm_flops = B * P * (d + 2 * d * A + 2 * A + 2 * d)
    report("afm_fwd(A=8)", B, graph_time(lambda: ops.afm_fwd(emb, W, bA, h, pw, pb)), B * (D + 1) * 4, flops=afm_flops)
    report("afm_bwd(A=8)", B, graph_time(lambda: ops.afm_bwd(emb, do, W, bA, h, pw)), B * (2 * D + 1) * 4, flops=3 * afm_flops)

    # FFM on field-aware rows of F*d floats; vocabularies as in bench.py (2.7 M rows: 6.8 GB, 20 GB with Adam's moments).
    # Two index distributions: skewed (hot rows, most reads served by L2 / Infinity Cache) and uniform (nearly every
    # row distinct).
    vocabs = [100] * 13 + [int(10 ** (1 + 5 * i / 25)) for i in range(26)]
    layout = ops.FieldLayout(vocabs, dev)
    T = torch.randn(layout.total_rows, F * d, device=dev) * 1e-2
    mT, vT = torch.zeros_like(T), torch.zeros_like(T)
    hyper = ops.make_hyper(dev, 1e-3)
    scal = torch.tensor([0.0, 1.0], device=dev)
    plan = ops.FFMPlan(layout, cap=B, d=d)
    parts = torch.zeros(plan.n_partials_wide, device=dev)
    for dist in ("skewed", "uniform"):
        u = torch.rand(F, B, generator=g)
        if dist == "skewed":
            u = u ** 6
        idx = torch.stack([(1 + u[i] * (v - 1)).long().clamp(1, v - 1) for i, v in enumerate(vocabs)], 0).to(dev)
        o = ops.ffm_fwd(idx, layout, T)
        do = torch.randn_like(o)
        uniq = sum(int(torch.unique(idx[i]).numel()) for i in range(F))
        fb = B * F * 8 + uniq * F * d * 4 + B * 4          # every distinct row from HBM once
        cb = B * F * (8 + F * d * 4) + B * 4               # what the samples read through the caches
        note = f"{dist} indices, {uniq} distinct rows of {F * d} floats among {layout.total_rows}"
        report("ffm_fwd", B, graph_time(lambda: ops.ffm_fwd(idx, layout, T), 5), fb, note=note, cache_bytes=cb)
        report("ffm_bwd_rows (sort + touched-row gradients)", B, graph_time(lambda: plan.rows(idx, T, do, parts), 5),
               fb + uniq * F * d * 4, note=note, cache_bytes=cb + uniq * F * d * 4)
        report("ffm touched-rows Adam", B, graph_time(lambda: plan.update_wide("Adam", hyper, scal, T, mT, vT), 5),
               uniq * F * d * 4 * 7, note=note + "; reads g,w,m,v and writes w,m,v")
    if B <= 4096:
        sp = ops.SparsePlan(layout, cap=B, d=1, with_lr=False)
        t = graph_time(lambda: ops.ffm_bwd(sp, idx, T, do), 3)
        report("ffm_bwd dense dT incl. zero fill (autograd path)", B, t, fb + 2 * layout.total_rows * F * d * 4)


if __name__ == "__main__":
    for B in ([int(a) for a in sys.argv[1:]] or [4096, 16384]):
        run(B)
