"""Dropout (model/utils/Layers.py:33-37,46; the reference's DeepFM script trains with dropout_rate=0.9 on the first
block, model/DeepFM/DeepFM.py:61) on the HIP kernel.  Its masks come from a counter-based generator, not from torch's,
so parity is checked with the mask made explicit: the kernel applied to a tensor of ones yields the multiplier
keep/(1-p) of every element, which the CPU oracle then applies at the same place."""
import pytest
import torch

from oracle import ctr_oracle as O

pytestmark = pytest.mark.gpu


def test_mask_statistics_and_determinism():
    from recommendsystem_amd import ops
    n = 1 << 20
    ones = torch.ones(n, device="cuda")
    for p in (0.1, 0.5, 0.9):
        m = ops.dropout_apply(ones, p, seed=2021, salt=3, call_no=7)
        vals = torch.unique(m)
        assert vals.numel() == 2 and float(vals[0]) == 0.0
        torch.testing.assert_close(vals[1].cpu(), torch.tensor(1.0 / (1.0 - p), dtype=torch.float32))
        keep = float((m != 0).float().mean())
        assert abs(keep - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5          # five standard deviations
        assert torch.equal(m, ops.dropout_apply(ones, p, seed=2021, salt=3, call_no=7))
        dev = torch.tensor([5], dtype=torch.int64, device="cuda")
        assert torch.equal(m, ops.dropout_apply(ones, p, seed=2021, salt=3, call_no=2, call_dev=dev))   # 2 + 5 = 7
        for other in (dict(seed=2022, salt=3, call_no=7), dict(seed=2021, salt=4, call_no=7), dict(seed=2021, salt=3, call_no=8)):
            assert not torch.equal(m, ops.dropout_apply(ones, p, **other))
    x = torch.randn(1001, device="cuda")                                    # ragged length, in place
    want = ops.dropout_apply(x, 0.3, 1, 1, 1)
    ops.dropout_apply(x, 0.3, 1, 1, 1, out=x)
    assert torch.equal(x, want)
    assert torch.equal(ops.dropout_apply(ones, 0.0, 1, 1, 1), ones)
    assert torch.count_nonzero(ops.dropout_apply(ones, 1.0, 1, 1, 1)).item() == 0


def test_block_with_dropout_forward_and_backward():
    from recommendsystem_amd import functional as Fn
    from recommendsystem_amd import layers, ops
    torch.manual_seed(11)
    blk = layers.MLPBlock(24, 40, "ReLU", None, dropout_rate=0.4).cuda().train()
    x = torch.randn(16, 24)
    g = torch.randn(16, 40)
    xg = x.cuda().requires_grad_(True)
    y = blk(xg)
    mask = ops.dropout_apply(torch.ones(16, 40, device="cuda"), 0.4, torch.initial_seed(), blk.dropout.salt,
                             Fn._dropout_calls).cpu()
    assert 0 < int((mask == 0).sum()) < mask.numel()
    y.backward(g.cuda())
    xr = x.clone().requires_grad_(True)
    W, b = blk.fc.weight.detach().cpu().requires_grad_(True), blk.fc.bias.detach().cpu().requires_grad_(True)
    yr = O.mlp_block(xr, W, b, "ReLU", None) * mask
    yr.backward(g)
    torch.testing.assert_close(y.detach().cpu(), yr.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(blk.fc.weight.grad.cpu(), W.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(blk.fc.bias.grad.cpu(), b.grad, rtol=1e-5, atol=1e-6)
    blk.eval()
    with torch.no_grad():                                                   # evaluation: identity
        torch.testing.assert_close(blk(x.cuda()).cpu(), O.mlp_block(x, W.detach(), b.detach(), "ReLU", None), rtol=1e-5, atol=1e-6)


def test_residual_block_with_dropout():
    """The reference drops the block's output before the residual add (Layers.py:88-94)."""
    from recommendsystem_amd import functional as Fn
    from recommendsystem_amd import layers, ops
    torch.manual_seed(5)
    for i, o in ((12, 12), (12, 7)):
        ref = O.OracleResidualBlock(i, o, "tanh", "ln").train()
        blk = layers.ResidualMLPBlock(i, o, "tanh", "ln", dropout_rate=0.5)
        blk.load_state_dict(ref.state_dict())
        blk = blk.cuda().train()
        x, g = torch.randn(9, i), torch.randn(9, o)
        xg = x.cuda().requires_grad_(True)
        y = blk(xg)
        mask = ops.dropout_apply(torch.ones(9, o, device="cuda"), 0.5, torch.initial_seed(), blk.dropout.salt,
                                 Fn._dropout_calls).cpu()
        y.backward(g.cuda())
        xr = x.clone().requires_grad_(True)
        out = O._activation("tanh", ref.norm_layer(ref.fc(xr))) * mask
        yr = (xr if ref.downsample is None else ref.downsample(xr)) + out
        yr.backward(g)
        torch.testing.assert_close(y.detach().cpu(), yr.detach(), rtol=1e-5, atol=2e-6)
        torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=1e-5, atol=2e-6)
        torch.testing.assert_close(blk.fc.weight.grad.cpu(), ref.fc.weight.grad, rtol=1e-5, atol=2e-6)


class _OracleWithMask(torch.nn.Module):
    """DeepFM of the oracle with a given multiplier on the first block's output (model/DeepFM/DeepFM.py:25-26: only the
    first MLPBlock gets the dropout rate)."""

    def __init__(self, o):
        super().__init__()
        self.o = o
        self.mask = None

    def forward(self, x):
        o = self.o
        e = o.embed(x)
        h = o.deep[0](e) * self.mask
        for blk in list(o.deep)[1:]:
            h = blk(h)
        return o.lr(x) + O.fm_second_order(e) + h


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("optimizer", ["SGD", "Adam"])
def test_fused_step_with_the_reference_scripts_dropout(graph, optimizer):
    from recommendsystem_amd import data, models, ops
    from recommendsystem_amd.base_model import random_all
    random_all(9)
    vocab, B, d, hidden, p = [30, 200, 7, 1000], 64, 8, [32, 16], 0.9
    rec = data.criteo_x1_recorder(embedding_dim=d, batch_size=B, vocab_sizes=vocab)
    m = models.DeepFM(rec, 1e-2, 0, hidden_dim_list=hidden, normalization=None, dropout_rate=p, optimizer_type=optimizer)
    assert m.deep[0].dropout.p == p and m.deep[1].dropout.p == 0
    with torch.no_grad():
        for n, q in m.named_parameters():
            if "embedded" in n:
                q[1:].normal_(0, 0.3)
    m.train()
    o = O.OracleCTRModel("deepfm", rec.feature_map, rec.features, d, hidden=hidden)
    sd = {}
    for k, v in m.state_dict().items():
        if k.startswith("embedding_layer.embedded_features."):
            sd["emb." + k[len("embedding_layer.embedded_features."):]] = v.cpu().clone()
        elif k.startswith("lr_layer.scalar.embedded_to_1_dim_features."):
            sd["emb1." + k[len("lr_layer.scalar.embedded_to_1_dim_features."):]] = v.cpu().clone()
        elif k == "lr_layer.bias":
            sd["bias"] = v.cpu().clone()
        elif k.startswith("deep."):
            sd[k] = v.cpu().clone()
    o.load_state_dict(sd)
    om = _OracleWithMask(o.train())
    opt = O.make_optimizer(optimizer, o.parameters(), 1e-2)
    eng = m.use_fused_step(mode="sparse", use_graph=graph)
    idxs, labs = data.synthetic_batches(vocab, B, 3, "cpu", seed=4)
    masks = []
    for step, (idx, y) in enumerate(zip(idxs, labs)):
        stepno = torch.tensor([step], dtype=torch.int64, device="cuda")
        om.mask = ops.dropout_apply(torch.ones(B, hidden[0], device="cuda"), p, torch.initial_seed(), m.deep[0].dropout.salt,
                                    0, stepno).cpu()
        masks.append(om.mask)
        loss = eng.step_tensors(idx.cuda(), y.cuda())
        ref, _ = O.train_step(om, opt, {c: idx[i] for i, c in enumerate(rec.features)}, y)
        torch.testing.assert_close(loss.cpu()[0], ref, rtol=2e-5, atol=1e-6)
    assert not torch.equal(masks[0], masks[1])                              # a fresh mask at every (replayed) step
    osd = o.state_dict()
    for k, v in m.state_dict().items():
        ok = ("emb." + k[len("embedding_layer.embedded_features."):] if k.startswith("embedding_layer.embedded_features.")
              else "emb1." + k[len("lr_layer.scalar.embedded_to_1_dim_features."):]
              if k.startswith("lr_layer.scalar.embedded_to_1_dim_features.") else "bias" if k == "lr_layer.bias"
              else k if k.startswith("deep.") else None)
        if ok is not None:
            torch.testing.assert_close(v.cpu(), osd[ok], rtol=1e-4, atol=5e-5 if optimizer == "Adam" else 5e-6,
                                       msg=lambda s, k=k: f"{k}: {s}")


def test_generic_trainer_draws_a_fresh_mask_at_every_graph_replay():
    """The touched-rows trainer replays the model's own forward from a HIP graph; the dropout masks are numbered by the
    trainer's device step counter, so replays of one captured step must not repeat a mask."""
    from recommendsystem_amd import data, models
    from recommendsystem_amd.base_model import random_all
    random_all(13)
    vocab, B = [30, 200, 7], 128
    rec = data.criteo_x1_recorder(embedding_dim=8, batch_size=B, vocab_sizes=vocab)
    m = models.PNN(rec, 0.0, 0, hidden_dim_list=[32, 16], pnn_type="inner", normalization=None, dropout_rate=0.5,
                   optimizer_type="SGD", activation_func="tanh")
    with torch.no_grad():
        for n, q in m.named_parameters():
            if "embedded" in n:
                q[1:].normal_(0, 0.3)
    m.train()
    eng = m.use_sparse_step(use_graph=True, lr=0.0)                       # lr = 0: the weights stay, only the masks change
    idxs, labs = data.synthetic_batches(vocab, B, 1, "cuda", seed=4)
    losses = [float(eng.step_tensors(idxs[0], labs[0], resident=True).item()) for _ in range(4)]
    assert len(set(losses)) == 4, losses
    m.eval()
    with torch.no_grad():
        a = m.logits({c: idxs[0][i] for i, c in enumerate(rec.features)})
        b = m.logits({c: idxs[0][i] for i, c in enumerate(rec.features)})
    assert torch.equal(a, b)                                               # evaluation: no dropout


@pytest.mark.parametrize("act", ["relu", "leakyrelu", "tanh"])
def test_stack_with_dropout_in_the_gemm_epilogues(act):
    """layers.MLPStack with dropout on its first two blocks: for relu / leakyrelu the mask rides in the forward GEMM's
    epilogue and, with act', in the next block's dX GEMM (rs_gemm_f32_drop); tanh takes the separate launches.  Either
    way the result is the oracle's with the same multipliers applied behind the blocks."""
    from recommendsystem_amd import functional as Fn
    from recommendsystem_amd import layers, ops
    torch.manual_seed(21)
    B = 320
    blocks = [layers.MLPBlock(36, 64, act, None, dropout_rate=0.5), layers.MLPBlock(64, 32, act, None, dropout_rate=0.25),
              layers.MLPBlock(32, 8, act, None)]
    stack = layers.MLPStack(*blocks).cuda().train()
    x = torch.randn(B, 36)
    g = torch.randn(B, 8)
    xg = x.cuda().requires_grad_(True)
    calls0 = Fn._dropout_calls
    y = stack(xg)
    seed = torch.initial_seed()
    masks = [ops.dropout_apply(torch.ones(B, w, device="cuda"), p, seed, blk.dropout.salt, calls0 + k + 1).cpu()
             for k, (blk, w, p) in enumerate(((blocks[0], 64, 0.5), (blocks[1], 32, 0.25)))]
    assert all(0 < int((m == 0).sum()) < m.numel() for m in masks)
    y.backward(g.cuda())
    xr = x.clone().requires_grad_(True)
    ps = [(b.fc.weight.detach().cpu().requires_grad_(True), b.fc.bias.detach().cpu().requires_grad_(True)) for b in blocks]
    h = O.mlp_block(xr, ps[0][0], ps[0][1], act, None) * masks[0]
    h = O.mlp_block(h, ps[1][0], ps[1][1], act, None) * masks[1]
    yr = O.mlp_block(h, ps[2][0], ps[2][1], act, None)
    yr.backward(g)
    torch.testing.assert_close(y.detach().cpu(), yr.detach(), rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=1e-5, atol=2e-6)
    for b, (W, bias) in zip(blocks, ps):
        torch.testing.assert_close(b.fc.weight.grad.cpu(), W.grad, rtol=1e-5, atol=5e-6)
        torch.testing.assert_close(b.fc.bias.grad.cpu(), bias.grad, rtol=1e-5, atol=5e-6)


def test_masks_do_not_depend_on_what_was_built_before():
    """Two models built alike, with another model (and stray layers) built in between, number their dropout layers
    alike -- by position in the model -- and so draw the same masks from the same seed: the same first training loss."""
    from recommendsystem_amd import data, layers, models
    from recommendsystem_amd.base_model import random_all
    vocab = [9, 30, 200]

    def make():
        random_all(11)
        rec = data.criteo_x1_recorder(embedding_dim=8, batch_size=32, vocab_sizes=vocab)
        m = models.DeepFM(rec, 1e-2, 0, hidden_dim_list=[16, 8], normalization=None, dropout_rate=0.5)
        m.train()
        return m

    a = make()
    layers.Dropout(0.3), layers.Dropout(0.1)
    models.DNN(data.criteo_x1_recorder(embedding_dim=4, batch_size=8, vocab_sizes=vocab), 1e-2, 0,
               hidden_dim_list=[8], normalization=None, dropout_rate=0.2)
    b = make()
    salts = lambda m: [mod.salt for mod in m.modules() if isinstance(mod, layers.Dropout)]
    assert salts(a) == salts(b) == list(range(1, len(salts(a)) + 1))
    idxs, labs = data.synthetic_batches(vocab, 32, 1, "cuda", seed=4)
    la = a.use_fused_step(mode="sparse").step_tensors(idxs[0], labs[0]).clone()
    lb = b.use_fused_step(mode="sparse").step_tensors(idxs[0], labs[0]).clone()
    assert torch.equal(la, lb)
