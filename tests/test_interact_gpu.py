"""K2 parity: feature-interaction kernels (PNN inner/outer, DCN cross, NFM, AFM, FFM) forward and backward
against the CPU oracle (oracle/ctr_oracle.py, autograd for the gradients) and against the vectors captured
from the reference's own methods (tests/golden/interactions.npz).  Tolerance: 1e-5 relative on fp32."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ctr_oracle as O

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6
DEV = "cuda:0"


def close(a, b, rtol=RTOL, atol=ATOL, msg=""):
    torch.testing.assert_close(a.detach().cpu(), b.detach().cpu(), rtol=rtol, atol=atol, msg=lambda s: f"{msg}: {s}")


def G(name):
    return torch.from_numpy(load_golden("interactions.npz")[name])


SHAPES = [(1, 2, 4), (6, 5, 4), (33, 7, 8), (64, 26, 16), (130, 39, 16), (2, 39, 16), (9, 3, 3), (5, 70, 4), (17, 4, 32)]


# ---------------------------------------------------------------- golden vectors (reference outputs)
def test_golden_pnn():
    from recommendsystem_amd import ops
    emb = G("emb").to(DEV)
    close(ops.pnn_inner_fwd(emb), G("pnn_inner"))
    close(ops.pnn_outer_fwd(emb), G("pnn_outer"))


def test_golden_dcn():
    from recommendsystem_amd import ops
    g = load_golden("interactions.npz")
    x0 = torch.from_numpy(g["emb"]).reshape(6, -1).to(DEV)
    w = torch.cat([torch.from_numpy(g[f"dcn.w.{l}"]).view(1, -1) for l in range(3)]).to(DEV)
    b = torch.cat([torch.from_numpy(g[f"dcn.b.{l}"]).view(1, -1) for l in range(3)]).to(DEV)
    out, _ = ops.dcn_cross_fwd(x0, w, b)
    close(out, G("dcn_out"), rtol=2e-5, atol=1e-5)


def test_golden_nfm():
    from recommendsystem_amd import ops
    E = G("emb").reshape(6, -1).to(DEV)
    out, _ = ops.nfm_bi_fwd(E, G("nfm_v").to(DEV))
    close(out, G("nfm_bi"), rtol=2e-5, atol=1e-5)


def test_golden_afm():
    from recommendsystem_amd import ops
    out = ops.afm_fwd(G("afm_emb").to(DEV), G("afm.att_w").to(DEV), G("afm.att_b").to(DEV), G("afm.att_h").to(DEV),
                      G("afm.p_w").to(DEV), G("afm.p_b").to(DEV))
    close(torch.sigmoid(G("lr").to(DEV) + out.view(-1, 1)), G("afm_out"))       # the fixture is sigmoid(lr + afm term)


def test_golden_ffm():
    from recommendsystem_amd import ops
    g = load_golden("interactions.npz")
    Fn, d, V = 5, 4, 9
    # T[row of feature i value x][j*d:(j+1)*d] = table_{i,j}[x]
    T = torch.cat([torch.cat([torch.from_numpy(g[f"ffm.table.{i}.{j}"]) for j in range(Fn)], 1) for i in range(Fn)], 0)
    layout = ops.FieldLayout([V] * Fn, DEV)
    idx = torch.from_numpy(g["xi"]).t().contiguous().to(DEV)
    close(ops.ffm_fwd(idx, layout, T.to(DEV)).view(-1, 1), G("ffm_out"))
    layout.check_indices()


# ---------------------------------------------------------------- oracle, forward + backward
@pytest.mark.parametrize("B,Fn,d", SHAPES)
def test_pnn_inner(B, Fn, d):
    from recommendsystem_amd import interactions as X
    g = torch.Generator().manual_seed(B * 100 + Fn)
    emb = torch.randn(B, Fn, d, generator=g)
    e_ref = emb.clone().requires_grad_(True)
    ref = O.pnn_inner(e_ref.view(B, -1), Fn, d)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    e = emb.to(DEV).requires_grad_(True)
    out = X.pnn_inner(e)
    # sums of d (forward) / F (backward) signed products: the absolute floor scales with the terms' magnitude
    close(out, ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max()), msg="fwd")
    out.backward(gout.to(DEV))
    close(e.grad, e_ref.grad, rtol=1e-5, atol=1e-6 * float(e_ref.grad.abs().max()), msg="demb")


@pytest.mark.parametrize("B,Fn,d", SHAPES)
def test_pnn_outer(B, Fn, d):
    from recommendsystem_amd import interactions as X
    g = torch.Generator().manual_seed(B * 100 + Fn)
    emb = torch.randn(B, Fn, d, generator=g)
    e_ref = emb.clone().requires_grad_(True)
    ref = O.pnn_outer(e_ref.view(B, -1), Fn, d)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    e = emb.to(DEV).requires_grad_(True)
    out = X.pnn_outer(e)
    assert torch.equal(out.cpu(), ref.detach())          # one multiply per element: bit-exact
    out.backward(gout.to(DEV))
    close(e.grad, e_ref.grad, rtol=1e-5, atol=1e-6 * float(e_ref.grad.abs().max()), msg="demb")


@pytest.mark.parametrize("B,D,L", [(1, 8, 1), (6, 20, 3), (700, 624, 3), (65, 416, 3), (33, 100, 5), (4, 1030, 2)])
def test_dcn_cross(B, D, L):
    from recommendsystem_amd import ops
    from recommendsystem_amd.interactions import _CrossFn
    g = torch.Generator().manual_seed(B + D)
    x0 = torch.randn(B, D, generator=g) * 0.3
    w = torch.randn(L, D, generator=g) * (1.0 / D ** 0.5)
    b = torch.randn(L, D, generator=g) * 0.1
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x0, w, b))
    ref = O.dcn_cross(xr, [wr[l] for l in range(L)], [br[l] for l in range(L)])
    gout = torch.randn(B, D, generator=g)
    ref.backward(gout)
    xg = x0.to(DEV).requires_grad_(True)
    # the 2L parameters one by one, as the module holds them: (D, 1) each; once as separate tensors, once as
    # neighbouring views of one buffer (what a trainer's flat parameter buffer gives: no concatenation)
    for flat in (False, True):
        if flat:
            buf = torch.cat([w.reshape(-1), b.reshape(-1)]).to(DEV)
            ps = [buf[i * D:(i + 1) * D].view(D, 1).requires_grad_(True) for i in range(2 * L)]
        else:
            ps = [t.to(DEV).clone().view(D, 1).requires_grad_(True) for t in list(w) + list(b)]
        xg.grad = None
        out = _CrossFn.apply(xg, L, *ps)
        close(out, ref, rtol=2e-5, atol=2e-6, msg="fwd")
        out.backward(gout.to(DEV))
        close(xg.grad, xr.grad, rtol=5e-5, atol=5e-6, msg="dx0")
        wgrad = torch.stack([p.grad.view(-1) for p in ps[:L]])
        bgrad = torch.stack([p.grad.view(-1) for p in ps[L:]])
        close(wgrad, wr.grad, rtol=5e-5, atol=2e-5 * max(1.0, B ** 0.5 / 8), msg="dw")
        close(bgrad, br.grad, rtol=5e-5, atol=2e-5 * max(1.0, B ** 0.5 / 8), msg="db")


def test_dcn_module_reference_shapes():
    """Parameters keep the reference's (D, 1) shapes and names; gradients reach each of them."""
    from recommendsystem_amd.interactions import CrossNetwork
    torch.manual_seed(3)
    net = CrossNetwork(24, 3).to(DEV)
    assert [tuple(p.shape) for p in net.cross_w] == [(24, 1)] * 3 and [tuple(p.shape) for p in net.cross_b] == [(24, 1)] * 3
    assert set(net.state_dict()) == {f"cross_{k}.{i}" for k in "wb" for i in range(3)}
    x = (torch.randn(10, 6, 4) * 0.2).to(DEV).requires_grad_(True)
    out = net(x)
    ref = O.dcn_cross(x.detach().cpu().view(10, -1), [p.detach().cpu() for p in net.cross_w],
                      [p.detach().cpu().view(-1) for p in net.cross_b])
    close(out, ref, rtol=2e-5, atol=2e-6)
    out.sum().backward()
    assert all(p.grad is not None and p.grad.shape == p.shape for p in net.parameters())


@pytest.mark.parametrize("B,D,K", [(1, 8, 1), (6, 20, 3), (700, 624, 4), (65, 416, 4), (33, 100, 16)])
def test_nfm_bi(B, D, K):
    from recommendsystem_amd import interactions as X
    g = torch.Generator().manual_seed(B + D + K)
    E = torch.randn(B, D, generator=g) * 0.5
    V = torch.randn(D, K, generator=g)
    Er, Vr = E.clone().requires_grad_(True), V.clone().requires_grad_(True)
    ref = O.nfm_bi_interaction_reference(Er, Vr)
    gout = torch.randn(B, K, generator=g)
    ref.backward(gout)
    Eg, Vg = E.to(DEV).requires_grad_(True), V.to(DEV).requires_grad_(True)
    out = X.nfm_bi_interaction(Eg, Vg)
    scale = float(ref.abs().max())
    close(out, ref, rtol=2e-5, atol=2e-6 * max(1.0, scale), msg="fwd")
    out.backward(gout.to(DEV))
    close(Eg.grad, Er.grad, rtol=5e-5, atol=1e-5 * max(1.0, float(Er.grad.abs().max())), msg="dE")
    close(Vg.grad, Vr.grad, rtol=5e-5, atol=1e-5 * max(1.0, float(Vr.grad.abs().max())), msg="dV")


@pytest.mark.parametrize("B,Fn,d,A", [(1, 2, 4, 4), (6, 5, 4, 8), (33, 7, 8, 8), (64, 26, 16, 8), (130, 39, 16, 8),
                                      (600, 39, 16, 16), (5, 70, 4, 4), (17, 4, 32, 8), (9, 50, 16, 8), (3, 64, 8, 4),
                                      (7, 48, 4, 16), (5, 16, 16, 4), (4, 17, 8, 16)])
def test_afm(B, Fn, d, A):
    from recommendsystem_amd.interactions import _AFMFn
    g = torch.Generator().manual_seed(B + Fn + d)
    emb = torch.randn(B, Fn, d, generator=g) * 0.7
    W = torch.randn(A, d, generator=g) * 0.5
    bA = torch.randn(A, generator=g) * 0.1
    h = torch.randn(1, A, generator=g)
    pw = torch.randn(1, d, generator=g)
    pb = torch.randn(1, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (emb, W, bA, h, pw, pb)]
    ref = O.afm_forward(*ref_in)
    gout = torch.randn(B, 1, generator=g)
    ref.backward(gout)
    ins = [t.to(DEV).requires_grad_(True) for t in (emb, W, bA, h, pw, pb)]
    out = _AFMFn.apply(*ins)
    close(out, ref, rtol=2e-5, atol=2e-6, msg="fwd")
    out.backward(gout.to(DEV))
    for name, a, r in zip(("demb", "dW", "dbA", "dh", "dpw", "dpb"), ins, ref_in):
        tol = 1e-5 * max(1.0, float(r.grad.abs().max()))
        close(a.grad, r.grad, rtol=1e-4, atol=tol, msg=name)


def test_afm_module_keeps_reference_names():
    from recommendsystem_amd.interactions import AttentionPooling
    m = AttentionPooling(8, 4)
    assert set(m.state_dict()) == {"attention.0.weight", "attention.0.bias", "attention.2.weight", "p.weight", "p.bias"}


@pytest.mark.parametrize("B,vocabs,d", [(6, [9] * 5, 4), (64, [11, 7, 50, 3, 200], 8), (300, [100] * 26, 16),
                                        (257, [10] * 13 + [int(10 ** (1 + 3 * i / 25)) for i in range(26)], 16),
                                        (3, [5, 6], 3)])
def test_ffm(B, vocabs, d):
    from recommendsystem_amd import ops
    g = torch.Generator().manual_seed(B + d)
    Fn = len(vocabs)
    tabs = [[torch.randn(v, d, generator=g) * 0.3 for _ in range(Fn)] for v in vocabs]
    # hot keys and padding entries
    x = torch.stack([(torch.rand(B, generator=g) ** 3 * v).long().clamp(0, v - 1) for v in vocabs], 1)
    ref_tabs = [[t.clone().requires_grad_(True) for t in row] for row in tabs]
    ref = O.ffm_interaction(ref_tabs, x)
    gout = torch.randn(B, 1, generator=g)
    ref.backward(gout)
    T = torch.cat([torch.cat(row, 1) for row in tabs], 0).to(DEV)
    layout = ops.FieldLayout(vocabs, DEV)
    idx = x.t().contiguous().to(DEV)
    out = ops.ffm_fwd(idx, layout, T)
    close(out.view(-1, 1), ref, rtol=2e-5, atol=2e-6 * max(1.0, float(ref.abs().max())), msg="fwd")
    plan = ops.SparsePlan(layout, cap=B, d=1, with_lr=False)
    dT = ops.ffm_bwd(plan, idx, T, gout.view(-1).to(DEV))
    want = torch.cat([torch.cat([t.grad if t.grad is not None else torch.zeros_like(t) for t in row], 1)
                      for row in ref_tabs], 0)
    # index 0 is the padding row: the reference's nn.Embedding(padding_idx=0) gives it no gradient
    o = 0
    for v in vocabs:
        want[o] = 0
        o += v
    close(dT, want, rtol=2e-5, atol=2e-6 * max(1.0, float(want.abs().max())), msg="dT")
    layout.check_indices()
    if d % 4 == 0:
        # touched-rows form: same sums in the same order, scattered back into a dense gradient; and its norm partials
        fp = ops.FFMPlan(layout, cap=B, d=d)
        parts = torch.full((fp.n_partials_wide,), float("nan"), device=DEV)
        fp.rows(idx, T, gout.view(-1).to(DEV), parts)
        dT2 = torch.zeros_like(T)
        fp.scatter_wide(dT2)
        # (rows spanning several chunks of the plan are summed chunk-wise: same terms, different association)
        close(dT2, dT, rtol=1e-5, atol=1e-6 * max(1.0, float(dT.abs().max())), msg="touched rows vs dense")
        close(parts.sum().cpu(), (dT2.double() ** 2).sum().float().cpu(), rtol=1e-5, atol=1e-12, msg="norm partials")
        # the row-sharded path works on gathered rows: same term, and its row gradients add up to the same dT
        flat = (idx + torch.tensor(layout.row_offsets, device=DEV).view(-1, 1)).t().contiguous()      # [B, F]
        Gr = T[flat]                                                                                   # [B, F, F*d]
        close(ops.ffm_rows_fwd(Gr, d), out, rtol=1e-6, atol=1e-7 * max(1.0, float(out.abs().max())), msg="rows fwd")
        dG = ops.ffm_rows_bwd(Gr, gout.view(-1).to(DEV), d)
        keep = (idx.t() > 0).reshape(-1)
        dT3 = torch.zeros_like(T).index_add_(0, flat.reshape(-1)[keep], dG.view(B * Fn, Fn * d)[keep])
        close(dT3, dT, rtol=1e-5, atol=1e-6 * max(1.0, float(dT.abs().max())), msg="rows bwd")
        wp = ops.WidePlan(layout, cap=B, W=Fn * d)
        wp.sort_columnar(idx)
        wparts = torch.full((wp.n_partials_wide,), float("nan"), device=DEV)
        wp.reduce_wide(wparts, dG.view(B * Fn, Fn * d))
        dT4 = torch.zeros_like(T)
        ops.call("rs_sparse_scatter_dense", ops.ptr(wp.plan), wp.F, wp.cap, wp.width, ops.ptr(wp.rowgrad_wide), None,
                 ops.ptr(dT4), None, ops.stream())
        close(dT4, dT, rtol=1e-5, atol=1e-6 * max(1.0, float(dT.abs().max())), msg="wide reduce")
        close(wparts.sum().cpu(), (dT4.double() ** 2).sum().float().cpu(), rtol=1e-5, atol=1e-12, msg="wide norm partials")
        # SGD on the touched rows == dense SGD
        hyper = ops.make_hyper(DEV, 0.1)
        scal = torch.tensor([0.0, 0.5], device=DEV)
        T2 = T.clone()
        fp.update_wide("SGD", hyper, scal, T2)
        close(T2, T - 0.1 * 0.5 * dT2, rtol=1e-6, atol=1e-7, msg="touched-rows SGD")


def test_unsupported_shapes_raise():
    from recommendsystem_amd import ops
    from recommendsystem_amd._lib import RSError
    with pytest.raises(RSError):
        ops.afm_fwd(torch.zeros(2, 3, 5, device=DEV), torch.zeros(4, 5, device=DEV), torch.zeros(4, device=DEV),
                    torch.zeros(4, device=DEV), torch.zeros(5, device=DEV), torch.zeros(1, device=DEV))
    with pytest.raises(RSError):
        ops.nfm_bi_fwd(torch.zeros(2, 8, device=DEV), torch.zeros(8, 17, device=DEV))


def test_wide_row_limits_raise():
    from recommendsystem_amd import ops
    from recommendsystem_amd._lib import RSError
    layout = ops.FieldLayout([5, 6, 7], DEV)
    with pytest.raises(RSError):
        ops.FFMPlan(layout, cap=8, d=3)                      # rows must be float4-divisible
    with pytest.raises(RSError):
        ops.FFMPlan(ops.FieldLayout([4] * 40, DEV), cap=8, d=32)   # 40 * 32 floats per row > 1024
    with pytest.raises(RSError):
        ops.WidePlan(layout, cap=8, W=1028)
    wp = ops.WidePlan(layout, cap=8, W=12)
    with pytest.raises(RSError):
        wp.reduce_wide(torch.zeros(wp.n_partials_wide, device=DEV), torch.zeros(24, 16, device=DEV))   # wrong width
    with pytest.raises(RSError):
        ops.ffm_rows_fwd(torch.zeros(2, 3, 10, device=DEV), 4)                                           # not F*d wide


@pytest.mark.parametrize("B,F,d", [(5, 7, 16), (3, 6, 4), (4, 5, 3), (2, 39, 16)])
def test_pnn_outer_pair_ranges(B, F, d):
    """rs_pnn_outer_{fwd,bwd}_range: the products of the pairs whose first field lies in a range, written into a block
    of columns of a wider buffer; laid end to end the ranges are the full product bit for bit, and their backward
    contributions add up to the full gradient."""
    from recommendsystem_amd import ops
    g = torch.Generator().manual_seed(B * F + d)
    emb = torch.randn(B, F, d, generator=g).to(DEV)
    full = ops.pnn_outer_fwd(emb)
    P, dd = F * (F - 1) // 2, d * d
    cuts = sorted({0, 1, F // 2, F - 2, F})
    wide = torch.full((B, P * dd + 8), float("nan"), device=DEV)
    for lo, hi in zip(cuts, cuts[1:]):
        c0, c1 = ops.pair_start(lo, F) * dd, ops.pair_start(hi, F) * dd
        if c1 > c0:
            ops.pnn_outer_fwd_range(emb, lo, hi, out=wide[:, c0:c1])
    assert torch.equal(wide[:, :P * dd], full)
    assert torch.isnan(wide[:, P * dd:]).all()
    dout = torch.randn(B, P * dd, generator=g).to(DEV)
    want = ops.pnn_outer_bwd(emb, dout)
    got = torch.empty_like(emb)
    first = True
    for lo, hi in zip(cuts, cuts[1:]):
        c0, c1 = ops.pair_start(lo, F) * dd, ops.pair_start(hi, F) * dd
        if c1 > c0:
            ops.pnn_outer_bwd_range(emb, dout[:, c0:c1], lo, hi, got, accumulate=not first)
            first = False
    close(got, want, rtol=1e-5, atol=1e-6 * float(want.abs().max()), msg="demb")


@pytest.mark.parametrize("with_other,norm,act", [(True, None, "relu"), (True, "bn", "tanh"), (False, None, "prelu"),
                                                 (True, "ln", "leakyrelu")])
def test_pnn_first_block_in_pair_blocks(with_other, norm, act, monkeypatch):
    """functional.pnn_outer_block (the first MLP layer of PNN over [x_other | outer products], a block of pairs at a
    time, weights in an aligned shadow: rs_pnn_outer_mlp_fwd / _bwd) against the same MLPBlock applied to the
    materialised concatenation (PNN.py:58), forward and every gradient; the block size is forced down so that the
    pairs fall into several blocks."""
    from recommendsystem_amd import _lib
    from recommendsystem_amd import functional as Fn
    from recommendsystem_amd import interactions as X
    from recommendsystem_amd import layers
    B, F, d, N = 96, 7, 4, 24
    monkeypatch.setattr(Fn, "OUTER_BLOCK_BYTES", B * d * d * 4 * 6)          # 6 pairs per block at most
    assert _lib.lib().rs_pnn_outer_mlp_blocks(B, F, d, Fn.OUTER_BLOCK_BYTES) >= 3
    assert _lib.lib().rs_pnn_outer_mlp_blocks(B, F, d, 0) == 1
    torch.manual_seed(11)
    K0 = F * d + 5 if with_other else 0                                      # an odd width in front of the products
    KO = F * (F - 1) // 2 * d * d
    blk = layers.MLPBlock(K0 + KO, N, act, norm).cuda().train()
    emb = (torch.randn(B, F, d, device=DEV) * 0.5).requires_grad_(True)
    other = torch.randn(B, K0, device=DEV).requires_grad_(True) if with_other else None
    gout = torch.randn(B, N, device=DEV)

    def run(fused):
        for p in blk.parameters():
            p.grad = None
        emb.grad = None
        if other is not None:
            other.grad = None
        if fused:
            y = Fn.pnn_outer_block(blk, emb, other)
        else:
            outer = X.pnn_outer(emb)
            y = blk(torch.cat([other, outer], 1) if other is not None else outer)
        (y * gout).sum().backward()
        grads = {k: p.grad.clone() for k, p in blk.named_parameters()}
        grads["emb"] = emb.grad.clone()
        if other is not None:
            grads["other"] = other.grad.clone()
        return y.detach(), grads

    y_ref, g_ref = run(False)
    y, g = run(True)
    close(y, y_ref, rtol=2e-5, atol=2e-5, msg="forward")
    for k in g_ref:
        atol = 1e-4 if (k == "fc.bias" and norm) else 2e-5 * max(1.0, float(g_ref[k].abs().max()))
        close(g[k], g_ref[k], rtol=1e-4, atol=atol, msg=k)


@pytest.mark.parametrize("B,Fn,d", [(3, 2, 4), (70, 7, 8), (300, 26, 16), (260, 39, 16), (33, 64, 4), (5, 8, 4), (9, 17, 32)])
def test_pnn_inner_input_rows(B, Fn, d):
    """functional.pnn_inner_input (the product kernel writes [embeddings | inner products | zero padding] itself,
    PNN.py:58,65-77) against cat([emb.flatten(1), oracle inner products]) and its gradient through both paths."""
    from recommendsystem_amd import functional as F_
    g = torch.Generator().manual_seed(B * 7 + Fn)
    emb = torch.randn(B, Fn, d, generator=g)
    K = Fn * d + Fn * (Fn - 1) // 2
    er = emb.clone().requires_grad_(True)
    ref = torch.cat([er.flatten(1), O.pnn_inner(er.flatten(1), Fn, d)], 1)
    gout = torch.randn(B, (K + 3) // 4 * 4, generator=g)
    (ref * gout[:, :K]).sum().backward()
    eg = emb.to(DEV).requires_grad_(True)
    x = F_.pnn_inner_input(eg)
    assert x is not None and x.shape == (B, (K + 3) // 4 * 4)
    assert torch.equal(x[:, :Fn * d].cpu(), emb.flatten(1))                  # the copy is exact
    assert (x[:, K:] == 0).all()
    close(x[:, Fn * d:K], ref[:, Fn * d:].detach(), rtol=1e-5, atol=1e-6 * float(ref.abs().max()), msg="products")
    (x * gout.to(DEV)).sum().backward()
    close(eg.grad, er.grad, rtol=1e-5, atol=1e-5 * float(er.grad.abs().max()), msg="demb")
