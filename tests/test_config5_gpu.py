"""The benchmark configurations at their own sizes against the CPU oracle (oracle/ctr_oracle.py: autograd +
clip_grad_norm_ + torch.optim from the same initial weights):

* PNN (inner), PNN (inner + outer) and DCN at the Criteo_x1 shape F = 39, d = 16 with the reference's literal tower
  [1024, 512, 256] (model/PNN/PNN.py:19-40, model/DCN/DCN.py:16-31) at B = 16384 -- the inner+outer case at a small
  batch, its first layer being 191,061 inputs wide;
* DeepFM with the tower [1024, 512, 256] (model/DeepFM/DeepFM.py:61) at B = 4096 under the reference's dense Adam.

Vocabularies are capped at 20,000 rows per field so that the oracle's dense optimizer stays quick on the host."""
import pytest
import torch

from oracle import ctr_oracle as O
from test_models_gpu import key_to_oracle

pytestmark = pytest.mark.gpu

VOCAB = [100] * 13 + [int(round(10 * (2e4 / 10) ** (i / 25))) for i in range(26)]
DIM = 16


def build(kind, optimizer, batch, lr=1e-2, **kw):
    from recommendsystem_amd import data, models
    from recommendsystem_amd.base_model import random_all
    random_all(5)
    rec = data.criteo_x1_recorder(embedding_dim=DIM, batch_size=batch, vocab_sizes=VOCAB)
    cls = {"pnn": models.PNN, "dcn": models.DCN, "deepfm": models.DeepFM}[kind]
    m = cls(rec, lr, 0, optimizer_type=optimizer, **kw)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embedded" in n:
                p[1:].normal_(0, 0.1)
            if n.startswith("cross."):
                p.mul_(0.04)                     # randn(D, 1) cross weights (DCN.py:16): keep x_l of order one at D = 624
    m.train()
    okw = {"hidden": kw["hidden_dim_list"], "normalization": kw.get("normalization")}
    for a, b in (("pnn_type", "pnn_type"), ("cross_layers_num", "cross_layers_num")):
        if a in kw:
            okw[b] = kw[a]
    o = O.OracleCTRModel(kind, rec.feature_map, rec.features, DIM, **okw)
    sd = {}
    for k, v in m.state_dict().items():
        ok = key_to_oracle(k)
        if ok is not None:
            sd[ok] = v.detach().cpu().clone()
    missing, unexpected = o.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k == "bias" or k.startswith("emb1.") for k in missing), missing
    o.train()
    return m, o, rec


def run_steps(kind, m, o, rec, optimizer, batch, steps, lr, driver, rtol=1e-4, atol_adam=5e-5, adam_grad_floor=0.0):
    from recommendsystem_amd import data
    opt = O.make_optimizer(optimizer, o.parameters(), lr)
    idxs, labs = data.synthetic_batches(VOCAB, batch, steps, "cpu", seed=9)
    for step, (idx, y) in enumerate(zip(idxs, labs)):
        x = {c: idx[i] for i, c in enumerate(rec.features)}
        if step == 0:
            with torch.no_grad():
                want = o(x)
                got = m.logits({c: v.cuda() for c, v in x.items()})
            # a logit is the end of sums over up to 191,061 products: the floor follows the largest logit, not each one
            scale = max(1.0, float(want.abs().max()))
            torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=5e-6 * scale, msg=lambda s: f"logits: {s}")
        loss = driver(idx, y)
        ref_loss, _ = O.train_step(o, opt, x, y)
        torch.testing.assert_close(loss.detach().cpu().view(-1)[0], ref_loss.view(()), rtol=2e-5, atol=1e-6)
    osd = o.state_dict()
    kept = total = 0
    for k, v in m.state_dict().items():
        ok = key_to_oracle(k)
        if ok is None or ok not in osd or not v.dtype.is_floating_point:
            continue
        atol = atol_adam if optimizer in ("Adam", "AdamW") else 5e-6
        got, want = v.cpu(), osd[ok]
        if adam_grad_floor:
            # one Adam step moves a weight by lr * g / (|g| + eps): where |g| is of the order of eps = 1e-8 the step is
            # decided by the rounding of g (whose summation order differs between the two implementations), so those
            # few elements are left out -- and must stay few
            g = dict(o.named_parameters())[ok].grad
            keep = (g.abs() >= adam_grad_floor) | (g == 0)          # exact zeros (untouched rows) are not noise
            kept += int(keep.sum())
            total += keep.numel()
            got, want = torch.where(keep, got, want), want
        torch.testing.assert_close(got, want, rtol=rtol, atol=atol, msg=lambda s, k=k: f"{k}: {s}")
    assert total == 0 or kept > 0.97 * total, (kept, total)
    m.embedding_layer._tab.layout.check_indices()


@pytest.mark.parametrize("kind,batch,kw", [
    ("pnn", 16384, dict(hidden_dim_list=[1024, 512, 256], pnn_type="inner", normalization=None)),
    ("pnn", 48, dict(hidden_dim_list=[1024, 512, 256], pnn_type="both", normalization=None)),
    ("dcn", 16384, dict(hidden_dim_list=[1024, 512, 256], cross_layers_num=3, normalization=None)),
    ("dcn", 16384, dict(hidden_dim_list=[1024, 512, 256], cross_layers_num=3, normalization="bn")),
], ids=["pnn-inner-16384", "pnn-both-48", "dcn-16384", "dcn-bn-16384"])
def test_config5_touched_rows_step(kind, batch, kw):
    """Two SGD steps of the touched-rows trainer (graph replay): losses and every parameter."""
    lr = 0.05
    m, o, rec = build(kind, "SGD", batch, lr=lr, **kw)
    eng = m.use_sparse_step(optimizer_type="SGD", lr=lr, use_graph=True)
    run_steps(kind, m, o, rec, "SGD", batch, 2, lr, lambda idx, y: eng.step_tensors(idx.cuda(), y.cuda()))
    assert eng.step_count.item() == 2


def test_pnn_inner_module_path_adam_at_16384():
    """BaseModel.train_step itself (autograd, clip_grad_norm_, torch.optim.Adam on dense table gradients)."""
    m, o, rec = build("pnn", "Adam", 16384, lr=1e-3, hidden_dim_list=[1024, 512, 256], pnn_type="inner", normalization=None)

    def driver(idx, y):
        x = {c: idx[i] for i, c in enumerate(rec.features)}
        return m.train_step(x, {"label": y})
    run_steps("pnn", m, o, rec, "Adam", 16384, 1, 1e-3, driver, adam_grad_floor=1e-7)


@pytest.mark.parametrize("graph", [False, True])
def test_deepfm_benchmark_tower_dense_adam(graph):
    """DeepFM 624 -> 1024 -> 512 -> 256 -> 1 at B = 4096: three steps of the fused engine in dense mode (the
    reference's optimizer semantics: every row of every table stepped by Adam) against the oracle."""
    m, o, rec = build("deepfm", "Adam", 4096, lr=1e-3, hidden_dim_list=[1024, 512, 256], normalization=None)
    eng = m.use_fused_step(mode="dense", use_graph=graph)
    # Adam turns a gradient at rounding level into a step of up to lr: the floor is a fraction of 3 * lr
    run_steps("deepfm", m, o, rec, "Adam", 4096, 3, 1e-3, lambda idx, y: eng.step_tensors(idx.cuda(), y.cuda()),
              rtol=1e-4, atol_adam=2e-5)


@pytest.mark.parametrize("opt", ["Adam", "AdamW", "SGD+reg"])
def test_benchmark_configuration_touched_rows_equals_dense_sweep(opt):
    """The benchmark's exact configuration -- 39 fields with the full vocabularies (2.71 M rows), embedding_dim 16, batch
    4096, tower 624-1024-512-256-1 -- for eight steps on non-repeating Zipf batches from one HIP graph per batch: the
    touched-rows step with replayed skips against the dense sweep over every row of every table (the reference's
    optimizer semantics, BaseModel.py:64-72,149), which needs no CPU oracle at this size.  Losses to 1e-5; weights to 2e-5
    with a floor of 1e-3 of one step (a rare row's first gradient can be as small as Adam's eps, where the step follows
    the gradient's rounding noise).

    "SGD+reg": SGD with regular = 1e-2 on tables of N(0, 0.05^2) rows (BaseModel.py:214-231: every one of the 2.71 M
    rows of every table, the model's unused twin tables included, shrinks by 1 - lr c lambda per step and counts in the
    clipped norm): the touched-rows step carries that in one scalar (ops.TableShrink) against the dense sweep."""
    from recommendsystem_amd import data, models
    from recommendsystem_amd.base_model import random_all
    vocab = data.criteo_x1_vocab_sizes()
    steps, lr = 8, 1e-3
    kw = {}
    if opt == "SGD+reg":
        opt, lr, kw = "SGD", 0.05, {"regular": 1e-2}
    idxs, labs = data.synthetic_batches(vocab, 4096, steps, "cuda", seed=77)
    out = []
    for mode in ("dense", "sparse"):
        random_all(3)
        rec = data.criteo_x1_recorder(embedding_dim=16, batch_size=4096, vocab_sizes=vocab)
        m = models.DeepFM(rec, lr, 0, hidden_dim_list=[1024, 512, 256], normalization=None, optimizer_type=opt, **kw)
        if kw:
            with torch.no_grad():
                for n, p in m.named_parameters():
                    if "embedded" in n:
                        p[1:].normal_(0, 0.05)
        m.train()
        eng = m.use_fused_step(mode=mode, use_graph=(mode == "sparse"))
        assert (eng.shrink is not None) == (bool(kw) and mode == "sparse")
        losses = [eng.step_tensors(idxs[s], labs[s], resident=True).clone() for s in range(steps)]
        eng.check()
        eng.flush()
        torch.cuda.synchronize()
        out.append((torch.cat(losses).cpu(), {k: v.detach().clone() for k, v in m.state_dict().items()}))
        del m, eng
    (ld, sd), (ls, ss) = out
    torch.testing.assert_close(ls, ld, rtol=1e-5, atol=1e-7)
    for k in sd:
        if sd[k].dtype.is_floating_point:
            torch.testing.assert_close(ss[k], sd[k], rtol=2e-5, atol=1e-6, msg=lambda s_, k=k: f"{k}: {s_}")
