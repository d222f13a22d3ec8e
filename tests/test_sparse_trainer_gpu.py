"""Touched-rows training step for every model kind (sparse_trainer.SparseTrainer) against the CPU oracle's dense
torch step (oracle/ctr_oracle.py: autograd + clip_grad_norm_ + torch.optim) from the same initial weights.

SGD and Adagrad touch only rows with a gradient, so sparse and dense steps agree for any number of steps; under Adam
the trainer replays a row's skipped zero-gradient steps before the row is next read (and state_dict() brings every
row up to date), so several steps agree as well."""
import pytest
import torch

from oracle import ctr_oracle as O
from test_models_gpu import BATCH, VOCAB, build, key_to_oracle, make_oracle, noise_driven

pytestmark = pytest.mark.gpu

CASES = [
    ("ffm", "SGD", 2, {}),
    ("ffm", "Adam", 3, {}),
    ("ffm", "Adagrad", 2, {}),
    ("pnn", "SGD", 2, dict(hidden_dim_list=[32, 16], pnn_type="both", normalization="bn")),
    ("pnn", "Adam", 3, dict(hidden_dim_list=[32, 16], pnn_type="inner", normalization=None)),
    ("nfm", "SGD", 2, dict(hidden_dim_list=[32, 16], hidden_dim=4, normalization="bn")),
    ("afm", "SGD", 2, dict(attention_dim=8)),
    ("afm", "Adagrad", 2, dict(attention_dim=4)),
    ("dcn", "SGD", 2, dict(hidden_dim_list=[32, 16], cross_layers_num=3, normalization="bn")),
    ("dcn", "Adam", 3, dict(hidden_dim_list=[32, 16], cross_layers_num=2, normalization=None)),
    ("afm", "AdamW", 3, dict(attention_dim=8)),
    ("widedeep", "SGD", 2, dict(hidden_dim_list=[32, 16], normalization="ln")),
    ("deepcrossing", "SGD", 2, dict(hidden_dim_list=[32, 16], normalization="bn")),
    ("deepcrossing", "Adam", 3, dict(hidden_dim_list=[32, 32], normalization="ln", activation_func="tanh")),
]


GRAPH_CASES = [CASES[0], CASES[1], CASES[3], CASES[6], CASES[8], CASES[9]]


@pytest.mark.parametrize("use_graph,kind,optimizer,steps,kw",
                         [(False, *c) for c in CASES] + [(True, *c) for c in GRAPH_CASES],
                         ids=[f"{c[0]}-{c[1]}-{i}" for i, c in enumerate(CASES)] +
                             [f"graph-{c[0]}-{c[1]}" for c in GRAPH_CASES])
def test_sparse_step_matches_dense_oracle(use_graph, kind, optimizer, steps, kw):
    # the reference's optimizer factory knows SGD / Adam / AdamW only: build with SGD, train with the requested one
    m, rec = build(kind, "SGD", **kw)
    o = make_oracle(kind, m, rec, kw)
    eng = m.use_sparse_step(optimizer_type=optimizer, lr=1e-2, use_graph=use_graph)
    opt = O.make_optimizer(optimizer, o.parameters(), 1e-2)
    g = torch.Generator().manual_seed(123)
    for step in range(steps):
        x = {c: (torch.rand(BATCH, generator=g) ** 2 * v).long().clamp(0, v - 1) for c, v in zip(rec.features, VOCAB)}
        y = (torch.rand(BATCH, generator=g) < 0.4).long()
        loss = m.train_step(x, {"label": y})
        ref_loss, _ = O.train_step(o, opt, x, y)
        torch.testing.assert_close(loss.cpu().view(()), ref_loss.view(()), rtol=2e-5, atol=1e-6)
        osd = o.state_dict()
        for k, v in m.state_dict().items():
            ok = key_to_oracle(k, kind)
            if kind == "nfm" and k == "linear.bias":
                ok = "linear.bias"
            if ok is None or not v.dtype.is_floating_point or noise_driven(kind, kw, k):
                continue
            atol = 5e-5 if optimizer in ("Adam", "AdamW", "Adagrad") else 5e-6
            torch.testing.assert_close(v.cpu(), osd[ok], rtol=1e-4, atol=atol, msg=lambda s, k=k: f"{k} after step {step}: {s}")
    assert eng.step_count.item() == steps
    if not use_graph and kind in ("pnn", "dcn", "nfm", "widedeep"):
        # the MLP blocks wrote their weight and bias gradients straight into the flat buffer
        assert eng.last["direct_grads"] >= 2 * len(kw["hidden_dim_list"])


def test_sparse_step_rejects_embedding_regulariser():
    from recommendsystem_amd._lib import RSError
    m, _ = build("afm", "SGD", attention_dim=4)
    m.emb_regular = 1e-4
    with pytest.raises(RSError):
        m.use_sparse_step()


@pytest.mark.parametrize("optimizer,use_graph", [("Adam", False), ("Adam", True), ("SGD", False)])
def test_ffm_step_at_criteo_shape(optimizer, use_graph):
    """The field-aware model at the benchmark's shape -- 39 fields, embedding_dim 16 (rows of 624 floats), 4096 samples
    -- with vocabularies cut to 5000 so that the CPU oracle's dense sweep stays short: rows come and go between the
    steps, so the wave-per-row replay of skipped Adam steps (rows wider than 256 floats) is what serves them."""
    from recommendsystem_amd import data, models
    from recommendsystem_amd.base_model import random_all
    random_all(5)
    vocab = [min(v, 5000) for v in data.criteo_x1_vocab_sizes()]
    B, dim, steps = 4096, 16, 4
    rec = data.criteo_x1_recorder(embedding_dim=dim, batch_size=B, vocab_sizes=vocab)
    m = models.FFM(rec, 1e-2, 0, optimizer_type="SGD")
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embedded" in n or "tables" in n:
                p[1:].normal_(0, 0.05)
    m.train()
    o = O.OracleCTRModel("ffm", rec.feature_map, rec.features, dim)
    sd = {key_to_oracle(k, "ffm"): v.detach().cpu().clone() for k, v in m.state_dict().items()
          if key_to_oracle(k, "ffm") is not None}
    missing, unexpected = o.load_state_dict(sd, strict=False)
    assert not unexpected and all(k == "bias" or k.startswith("emb1.") for k in missing), (missing, unexpected)
    o.train()
    eng = m.use_sparse_step(optimizer_type=optimizer, lr=1e-2, use_graph=use_graph)
    opt = O.make_optimizer(optimizer, o.parameters(), 1e-2)
    g = torch.Generator().manual_seed(321)
    for step in range(steps):
        x = {c: (torch.rand(B, generator=g) ** 3 * v).long().clamp(0, v - 1) for c, v in zip(rec.features, vocab)}
        y = (torch.rand(B, generator=g) < 0.4).long()
        loss = m.train_step(x, {"label": y})
        ref_loss, _ = O.train_step(o, opt, x, y)
        torch.testing.assert_close(loss.cpu().view(()), ref_loss.view(()), rtol=2e-5, atol=1e-6)
    osd = o.state_dict()
    for k, v in m.state_dict().items():
        ok = key_to_oracle(k, "ffm")
        if ok is None or not v.dtype.is_floating_point:
            continue
        if optimizer != "Adam":
            torch.testing.assert_close(v.cpu(), osd[ok], rtol=1e-4, atol=5e-6, msg=lambda s, k=k: f"{k}: {s}")
            continue
        # Adam: |diff| <= 5e-5 + 1e-4 |ref| but for a handful of elements in millions whose gradient is a cancelled
        # sum as small as eps (1e-8): there lr * m / (sqrt(v) + eps) follows g's rounding noise; those stay within 3 %
        # of one step (lr = 1e-2), where a wrong or missing replay is off by whole steps
        diff, ref = (v.cpu() - osd[ok]).abs(), osd[ok].abs()
        loose = diff > 5e-5 + 1e-4 * ref
        assert int(loose.sum()) <= max(1, diff.numel() // 200000), f"{k}: {int(loose.sum())} of {diff.numel()} elements off"
        assert float(diff.max()) <= 3e-4, f"{k}: greatest difference {float(diff.max())}"
    assert eng.step_count.item() == steps


@pytest.mark.parametrize("mode", ["sparse", "dense"])
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("optimizer,kw", [
    ("SGD", dict(hidden_dim_list=[32, 16], cross_layers_num=3, normalization=None)),
    ("Adam", dict(hidden_dim_list=[32, 16], cross_layers_num=2, normalization=None)),
    ("AdamW", dict(hidden_dim_list=[32, 16], cross_layers_num=3, normalization=None, activation_func="tanh")),
    ("SGD", dict(hidden_dim_list=[32, 16], cross_layers_num=2, normalization="bn")),
    ("Adam", dict(hidden_dim_list=[32], cross_layers_num=1, normalization="ln")),
], ids=["sgd", "adam", "adamw-tanh", "sgd-bn", "adam-ln"])
def test_dcn_on_the_fused_engine_matches_dense_oracle(optimizer, kw, graph, mode):
    """DCN (model/DCN/DCN.py:48-52) on the hand-scheduled step (engine.FusedTrainer: cross network forward / backward,
    the two-block 1-wide head with the tower's last act' fused, everything else as for DeepFM) against the CPU oracle's
    dense torch step, as the generic trainer is checked above."""
    m, rec = build("dcn", "SGD", **kw)
    o = make_oracle("dcn", m, rec, kw)
    eng = m.use_fused_step(mode=mode, optimizer_type=optimizer, lr=1e-2, use_graph=graph)
    opt = O.make_optimizer(optimizer, o.parameters(), 1e-2)
    g = torch.Generator().manual_seed(123)
    steps = 3
    for step in range(steps):
        x = {c: (torch.rand(BATCH, generator=g) ** 2 * v).long().clamp(0, v - 1) for c, v in zip(rec.features, VOCAB)}
        y = (torch.rand(BATCH, generator=g) < 0.4).long()
        loss = m.train_step(x, {"label": y})
        ref_loss, _ = O.train_step(o, opt, x, y)
        torch.testing.assert_close(loss.cpu().view(()), ref_loss.view(()), rtol=2e-5, atol=1e-6)
        osd = o.state_dict()
        for k, v in m.state_dict().items():
            ok = key_to_oracle(k, "dcn")
            if ok is None or not v.dtype.is_floating_point or noise_driven("dcn", kw, k):
                continue
            if kw.get("normalization") == "bn" and k.endswith(".fc.bias") and k.startswith("deep."):
                continue            # in front of a BatchNorm: gradient exactly zero, noise on both sides
            atol = 5e-5 if optimizer in ("Adam", "AdamW") else 5e-6
            torch.testing.assert_close(v.cpu(), osd[ok], rtol=1e-4, atol=atol, msg=lambda s, k=k: f"{k} after step {step}: {s}")
    assert eng.step_count.item() == steps


@pytest.mark.parametrize("mode", ["sparse", "dense"])
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("optimizer,kw", [
    ("SGD", dict(hidden_dim_list=[32, 16], pnn_type="inner", normalization=None)),
    ("Adam", dict(hidden_dim_list=[32, 16], pnn_type="inner", normalization=None)),
    ("AdamW", dict(hidden_dim_list=[32, 16], pnn_type="inner", normalization=None, activation_func="tanh")),
    ("SGD", dict(hidden_dim_list=[32, 16], pnn_type="inner", normalization="bn")),
    ("Adam", dict(hidden_dim_list=[32], pnn_type="inner", normalization="ln")),
], ids=["sgd", "adam", "adamw-tanh", "sgd-bn", "adam-ln"])
def test_pnn_inner_on_the_fused_engine_matches_dense_oracle(optimizer, kw, graph, mode):
    """PNN with inner products (model/PNN/PNN.py:58,80-89) on the hand-scheduled step (engine.FusedTrainer): the tower
    reads [embeddings | inner products] rows of 63 floats here, kept padded to 64 in the flat weight buffer, so the
    first layer's weight is a strided view of its padded block.  Checked against the CPU oracle's dense torch step."""
    m, rec = build("pnn", "SGD", **kw)
    o = make_oracle("pnn", m, rec, kw)
    eng = m.use_fused_step(mode=mode, optimizer_type=optimizer, lr=1e-2, use_graph=graph)
    w0 = m.mlp[0].fc.weight
    assert w0.shape[1] == 63 and w0.stride(0) == 64
    opt = O.make_optimizer(optimizer, o.parameters(), 1e-2)
    g = torch.Generator().manual_seed(123)
    steps = 3
    for step in range(steps):
        x = {c: (torch.rand(BATCH, generator=g) ** 2 * v).long().clamp(0, v - 1) for c, v in zip(rec.features, VOCAB)}
        y = (torch.rand(BATCH, generator=g) < 0.4).long()
        loss = m.train_step(x, {"label": y})
        ref_loss, _ = O.train_step(o, opt, x, y)
        torch.testing.assert_close(loss.cpu().view(()), ref_loss.view(()), rtol=2e-5, atol=1e-6)
        osd = o.state_dict()
        for k, v in m.state_dict().items():
            ok = key_to_oracle(k, "pnn")
            if ok is None or not v.dtype.is_floating_point:
                continue
            if kw.get("normalization") == "bn" and k.endswith(".fc.bias"):
                continue            # in front of a BatchNorm: gradient exactly zero, noise on both sides
            atol = 5e-5 if optimizer in ("Adam", "AdamW") else 5e-6
            torch.testing.assert_close(v.cpu(), osd[ok], rtol=1e-4, atol=atol, msg=lambda s, k=k: f"{k} after step {step}: {s}")
    assert eng.step_count.item() == steps
    # the pad column of the first layer's block stays zero under every optimizer
    full = eng._wfull[id(w0)]
    assert float(full[:, 63:].abs().max()) == 0.0
    # evaluation runs the model's own forward on the strided weight
    m.eval()
    o.eval()
    with torch.no_grad():
        torch.testing.assert_close(m.logits(x).cpu().view(-1), o(x).view(-1), rtol=1e-4, atol=1e-5)


def test_fused_engine_refuses_other_pnn_variants():
    from recommendsystem_amd._lib import RSError
    m, _ = build("pnn", "SGD", hidden_dim_list=[32, 16], pnn_type="outer", normalization=None)
    with pytest.raises(RSError):
        m.use_fused_step(mode="sparse")
