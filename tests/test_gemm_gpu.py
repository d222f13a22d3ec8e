"""K2e parity: fp32 MFMA GEMM (+ fused MLP epilogues) vs a float64 CPU product."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ACT = {"none": 0, "relu": 1, "tanh": 2, "leakyrelu": 3}


def _ref_act(code, z):
    if code == 1:
        return torch.relu(z)
    if code == 2:
        return torch.tanh(z)
    if code == 3:
        return torch.nn.functional.leaky_relu(z, 0.01)
    return z


def _check(got, want64, bound64, rel=2e-6):
    # |err| <= rel * sum_k |a||b|  (fp32 accumulation bound), elementwise
    err = (got.double().cpu() - want64).abs()
    assert (err <= rel * bound64 + 1e-30).all(), float((err / (bound64 + 1e-30)).max())


SHAPES = [(1, 1, 1), (3, 5, 7), (64, 64, 32), (65, 63, 33), (128, 128, 64), (200, 1, 256), (256, 624, 1024),
          (4096, 1024, 624), (512, 256, 512), (130, 260, 36), (31, 520, 14)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_layouts(M, N, K, layout):
    from recommendsystem_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g)          # logical B(k, n)
    want = A.double() @ Bm.double()
    bound = A.double().abs() @ Bm.double().abs()
    dev = "cuda:0"
    if layout == "nt":      # forward: B stored [N, K]
        got = ops.gemm(A.to(dev), Bm.t().contiguous().to(dev))
    elif layout == "nn":    # dX: B stored [K, N]
        got = ops.gemm(A.to(dev), Bm.to(dev), b_nc=True)
    else:                   # dW: A stored [K, M], B stored [K, N]
        got = ops.gemm(A.t().contiguous().to(dev), Bm.to(dev), a_mc=True, b_nc=True)
    _check(got, want, bound)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 5])
@pytest.mark.parametrize("splitk", [1, 3, 8])
def test_gemm_tiles_and_splitk(tile, splitk):
    from recommendsystem_amd import ops
    g = torch.Generator().manual_seed(tile * 10 + splitk)
    M, N, K = 300, 200, 1000
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    want = A.double() @ W.double().t()
    bound = A.double().abs() @ W.double().abs().t()
    got = ops.gemm(A.cuda(), W.cuda(), tile=tile, splitk=splitk)
    _check(got, want, bound)
    # split-K slabs are summed in a fixed order: bitwise reproducible
    again = ops.gemm(A.cuda(), W.cuda(), tile=tile, splitk=splitk)
    assert torch.equal(got, again)
    AT = A.t().contiguous().cuda()
    got = ops.gemm(AT, W.t().contiguous().cuda(), a_mc=True, b_nc=True, tile=tile, splitk=splitk)
    _check(got, want, bound)


@pytest.mark.parametrize("act", ["none", "relu", "tanh", "leakyrelu"])
def test_gemm_forward_epilogue(act):
    from recommendsystem_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 257, 130, 96
    X = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    want = _ref_act(ACT[act], X.double() @ W.double().t() + b.double())
    got = ops.gemm(X.cuda(), W.cuda(), bias=b.cuda(), act=ACT[act])
    torch.testing.assert_close(got.cpu().double(), want, rtol=1e-5, atol=1e-5)
    got = ops.gemm(X.cuda(), W.cuda(), bias=b.cuda(), act=ACT[act], splitk=2)
    torch.testing.assert_close(got.cpu().double(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("act", ["relu", "tanh", "leakyrelu"])
def test_gemm_backward_epilogue(act):
    """dZ_prev = (dZ W) * act'(H_prev), act' evaluated from the stored activation output."""
    from recommendsystem_amd import ops
    g = torch.Generator().manual_seed(9)
    M, N, K = 130, 72, 50          # dZ [M, N], W [N, K], H_prev [M, K]
    dZ = torch.randn(M, N, generator=g)
    W = torch.randn(N, K, generator=g)
    Zp = torch.randn(M, K, generator=g)
    Hp = _ref_act(ACT[act], Zp)
    if act == "relu":
        dact = (Zp > 0).double()
    elif act == "tanh":
        dact = 1 - torch.tanh(Zp.double()) ** 2
    else:
        dact = torch.where(Zp > 0, 1.0, 0.01).double()
    want = (dZ.double() @ W.double()) * dact
    got = ops.gemm(dZ.cuda(), W.cuda(), b_nc=True, mulsrc=Hp.cuda(), mul_act=ACT[act])
    torch.testing.assert_close(got.cpu().double(), want, rtol=1e-5, atol=1e-5)


def test_colsum():
    from recommendsystem_amd import ops
    g = torch.Generator().manual_seed(1)
    for M, N in [(1, 1), (7, 5), (4096, 1024), (1000, 130), (300, 1)]:
        X = torch.randn(M, N, generator=g)
        got = ops.colsum(X.cuda())
        torch.testing.assert_close(got.cpu().double(), X.double().sum(0), rtol=1e-6, atol=1e-4 * (M ** 0.5) * 1e-1)


def test_mfma_fragment_layout_with_asymmetric_data():
    """A = I against an asymmetric B catches a transposed C write or swapped operand lanes."""
    from recommendsystem_amd import ops
    n = 96
    Bm = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) + torch.arange(n).float()[:, None] * 0.5
    I = torch.eye(n)
    got = ops.gemm(I.cuda(), Bm.t().contiguous().cuda())
    assert torch.equal(got.cpu(), Bm)
    got = ops.gemm(I.cuda(), Bm.cuda(), b_nc=True)
    assert torch.equal(got.cpu(), Bm)
    got = ops.gemm(I.cuda(), Bm.cuda(), a_mc=True, b_nc=True)
    assert torch.equal(got.cpu(), Bm)


@pytest.mark.parametrize("M,N,K", [(4096, 1024, 512), (4096, 512, 256), (300, 200, 64), (64, 64, 32), (130, 70, 40),
                                   (1000, 37, 128), (33, 5, 7)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
def test_gemm_colsum_byproduct(M, N, K, tile):
    """Column sums of the stored C (after the epilogue) as a by-product, e.g. db of the dZ a dX-GEMM produces;
    several producers are finished by one launch."""
    from recommendsystem_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = torch.randn(K, N, generator=g).cuda()
    H = torch.randn(M, N, generator=g).cuda()
    cs = torch.full((N,), 7.0, device="cuda")
    cs_b = torch.full((N,), 7.0, device="cuda")
    parts = ops.ColsumParts(ops.Workspace("cuda:0"))
    for _ in range(2):
        C = ops.gemm(A, W, b_nc=True, mulsrc=H, mul_act=1, tile=tile, splitk=1, colsum_part=parts.expect(cs))
        C2 = ops.gemm(A, W, b_nc=True, tile=tile, splitk=1, colsum_part=parts.expect(cs_b))
        parts.finish()
        for c, s_ in ((C, cs), (C2, cs_b)):
            want = c.double().sum(0)
            bound = c.double().abs().sum(0)
            assert ((s_.double() - want).abs() <= 1e-6 * bound + 1e-30).all()
    plain = ops.gemm(A, W, b_nc=True, mulsrc=H, mul_act=1, tile=tile, splitk=1)
    assert torch.equal(plain, C)
    # split reductions fall back to a separate column-sum launch
    if K >= 64:
        cs2 = torch.zeros(N, device="cuda")
        ops.gemm(A, W, b_nc=True, tile=tile, splitk=2, colsum_part=parts.expect(cs2))
        parts.finish()
        ref = ops.gemm(A, W, b_nc=True, tile=tile, splitk=2).double()
        assert ((cs2.double() - ref.sum(0)).abs() <= 1e-6 * ref.abs().sum(0) + 1e-30).all()


def test_deferred_splitk_products_reduce_in_one_launch():
    """Weight-gradient shaped products leave their slabs behind; one rs_splitk_reduce_multi finishes all of them with
    the bits of the immediate form (same slabs, same slab order)."""
    from recommendsystem_amd import ops
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(3)
    shapes = [(256, 96, 4096), (64, 130, 2048), (33, 7, 4096)]        # (M, N, K); the last one has unaligned outputs
    ws = ops.Workspace(torch.device(DEV))
    coll = ops.SplitKParts(ws)
    outs, wants = [], []
    for M, N, K in shapes:
        a = torch.randn(K, M, generator=g).to(DEV)
        b = torch.randn(K, N, generator=g).to(DEV)
        wants.append(ops.gemm(a, b, a_mc=True, b_nc=True, tile=3, splitk=4))
        out = torch.full((M, N), float("nan"), device=DEV)
        ops.gemm(a, b, a_mc=True, b_nc=True, tile=3, splitk=4, out=out, defer_reduce=coll)
        outs.append(out)
    assert all(torch.isnan(o).all() for o in outs)                    # nothing written before finish()
    coll.finish()
    for o, w in zip(outs, wants):
        assert torch.equal(o, w)
    # a product that does not split is computed at once
    a, b = torch.randn(64, 32, generator=g).to(DEV), torch.randn(64, 48, generator=g).to(DEV)
    out = ops.gemm(a, b, a_mc=True, b_nc=True, tile=3, splitk=1, defer_reduce=coll)
    assert torch.equal(out, ops.gemm(a, b, a_mc=True, b_nc=True, tile=3, splitk=1))
    coll.finish()


@pytest.mark.parametrize("M,N,K,pad", [(256, 128, 96, 8), (200, 70, 37, 3), (4096, 1024, 512, 4), (130, 64, 1024, 0)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_column_blocks_and_running_sum(M, N, K, pad, layout):
    """Operands and result that are blocks of columns of wider matrices (row stride > row length), and
    rs_gemm_f32_acc: C = act(C + A B + bias) summed over two halves of the reduction dimension."""
    from recommendsystem_amd import ops
    g = torch.Generator().manual_seed(M + N + K + pad)
    dev = "cuda:0"
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    want = torch.tanh(A.double() @ Bm.double() + bias.double())
    bound = A.double().abs() @ Bm.double().abs() + bias.double().abs()

    def block(t):                        # t as columns [pad, pad + width) of a wider matrix
        wide = torch.full((t.shape[0], t.shape[1] + 2 * pad + 4), float("nan"))
        wide[:, pad:pad + t.shape[1]] = t
        return wide.to(dev)[:, pad:pad + t.shape[1]]

    kh = K // 2
    out = block(torch.zeros(M, N))
    for k0, k1, last in ((0, kh, False), (kh, K, True)):
        if layout == "nt":
            a, b, kw = block(A[:, k0:k1]), block(Bm[k0:k1].t().contiguous()), {}
        elif layout == "nn":
            a, b, kw = block(A[:, k0:k1]), block(Bm[k0:k1]), dict(b_nc=True)
        else:
            a, b, kw = block(A[:, k0:k1].t().contiguous()), block(Bm[k0:k1]), dict(a_mc=True, b_nc=True)
        ops.gemm(a, b, out=out, accumulate=k0 > 0, bias=bias.to(dev) if last else None, act=ACT["tanh"] if last else 0, **kw)
    _check(out, want, bound, rel=4e-6)


@pytest.mark.parametrize("stages", [2, 3])
def test_grouped_products_equal_single_launches(stages):
    """rs_gemm_f32_group: a layer's dW (split, slabs reduced later), its dX with the act' factor and the bias-gradient
    by-product, a forward product with bias + activation and a ragged unsplit dW in ONE launch give the bits of the
    four single launches."""
    from recommendsystem_amd import ops
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(11)
    B, n_in, n_out = 1024, 200, 136
    dz = torch.randn(B, n_out, generator=g).to(DEV)
    x = torch.randn(B, n_in, generator=g).to(DEV)
    W = torch.randn(n_out, n_in, generator=g).to(DEV)
    hprev = torch.randn(B, n_in, generator=g).to(DEV)
    bias = torch.randn(n_out, generator=g).to(DEV)
    xs = torch.randn(100, 36, generator=g).to(DEV)
    dzs = torch.randn(100, 52, generator=g).to(DEV)

    ws = ops.Workspace(torch.device(DEV))
    # single launches
    sk1 = ops.SplitKParts(ws)
    dW1 = torch.empty(n_out, n_in, device=DEV)
    ops.gemm(dz, x, a_mc=True, b_nc=True, out=dW1, tile=3, splitk=4, defer_reduce=sk1)
    sk1.finish()
    cs1 = ops.ColsumParts(ws, "one")
    db1 = torch.zeros(n_in, device=DEV)
    cs1.expect(db1)
    dX1 = ops.gemm(dz, W, b_nc=True, mulsrc=hprev, mul_act=1, tile=5, splitk=1, colsum_part=cs1)
    cs1.finish()
    H1 = ops.gemm(x, W, bias=bias, act=2, tile=5, splitk=1)
    dWs1 = ops.gemm(dzs, xs, a_mc=True, b_nc=True, tile=3, splitk=1)

    # one launch
    grp = ops.GemmGroup(stages)
    sk2 = ops.SplitKParts(ws)
    dW2 = torch.full((n_out, n_in), float("nan"), device=DEV)
    grp.add(dz, x, a_mc=True, b_nc=True, out=dW2, splitk=4, defer_reduce=sk2)
    cs2 = ops.ColsumParts(ws, "two")
    db2 = torch.zeros(n_in, device=DEV)
    cs2.expect(db2)
    dX2 = grp.add(dz, W, b_nc=True, mulsrc=hprev, mul_act=1, splitk=1, colsum_part=cs2)
    H2 = grp.add(x, W, bias=bias, act=2, splitk=1)
    dWs2 = grp.add(dzs, xs, a_mc=True, b_nc=True, splitk=1)
    assert len(grp) == 4
    grp.launch()
    sk2.finish()
    cs2.finish()
    assert torch.equal(dW2, dW1)
    assert torch.equal(dX2, dX1)
    assert torch.equal(db2, db1)
    assert torch.equal(H2, H1)
    assert torch.equal(dWs2, dWs1)
    # a dropout mask in the dX epilogue (mask regenerated from the key, then the act' factor): five products = two launches
    key = (0.5, 1234, 7, 3, torch.full((1,), 2, dtype=torch.int64, device=DEV))
    dXm1 = ops.gemm(dz, W, b_nc=True, mulsrc=hprev, mul_act=1, tile=5, splitk=1, dropout=key)
    grp.add(dz, x, a_mc=True, b_nc=True, out=dW2, splitk=4, defer_reduce=sk2)
    dXm2 = grp.add(dz, W, b_nc=True, mulsrc=hprev, mul_act=1, splitk=1, dropout=key)
    grp.add(x, W, bias=bias, act=2, splitk=1)
    grp.add(dzs, xs, a_mc=True, b_nc=True, splitk=1)
    H3 = grp.add(x, W, bias=bias, act=1, splitk=1, dropout=key)
    assert len(grp) == 5
    grp.launch()
    sk2.finish()
    assert torch.equal(dXm2, dXm1) and bool((dXm2 == 0).any()) and not torch.equal(dXm2, dX1)
    assert torch.equal(H3, ops.gemm(x, W, bias=bias, act=1, tile=5, splitk=1, dropout=key))
    # an operand that does not allow 16-byte accesses goes through a launch of its own
    a_odd = torch.randn(64, 33, generator=g).to(DEV)
    b_odd = torch.randn(48, 33, generator=g).to(DEV)
    o = grp.add(a_odd, b_odd)
    assert len(grp) == 0 and torch.equal(o, ops.gemm(a_odd, b_odd))
