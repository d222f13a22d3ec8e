"""End-to-end parity of the training step against vectors captured from the reference's
BaseModel.train_step (tests/golden/train_*.npz), for both drivers:

* the module path: reference-shaped nn.Modules + torch autograd + torch optimizers (BaseModel.train_step)
* the fused engine (engine.FusedTrainer), dense mode (exact reference semantics) and sparse mode.
"""
import json
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import golden_batch, golden_feature_map, load_golden

pytestmark = pytest.mark.gpu

FIXTURES = ["train_lr_adam_reg.npz", "train_fm_sgd.npz", "train_deepfm_adam.npz", "train_deepfm_sgd_bn.npz",
            "train_deepfm_adamw_reg.npz"]
# residual stacks (model/DeepCrossing/DeepCrossing.py:15-36): not part of the fused engine's model family
RESIDUAL_FIXTURES = ["train_deepcrossing_sgd_bn.npz", "train_deepcrossing_adam_ln.npz"]
LONG_FIXTURES = ["train_deepfm_adam_long.npz", "train_deepfm_adamw_long.npz"]
# SGD with emb_regular (BaseModel.py:214-231): every table row shrinks in every step; in the third trace lambda is
# large enough for clip_grad_norm_ to bite, so the shrink factor changes from step to step
REG_SGD_FIXTURES = ["train_lr_sgd_reg.npz", "train_deepfm_sgd_reg.npz", "train_fm_sgd_reg_clip.npz"]


def build_model(g):
    from recommendsystem_amd import models
    fm, feats = golden_feature_map(g)
    dim = int(g["dim"])
    n_cat = sum(1 for c in feats if fm[c]["type"] == "categorical")
    dr = SimpleNamespace(embedding_dim=dim, feature_map=fm, features=feats, feature_num=len(feats),
                         input_dim=n_cat * dim, batch_size=16)
    kind = str(g["kind"])
    reg = float(g["regular"])
    reg = None if reg < 0 else reg
    kw = dict(regular=reg, optimizer_type=str(g["optimizer"]))
    if kind == "lr":
        m = models.LR(dr, float(g["lr"]), 0, **kw)
    elif kind == "fm":
        m = models.FM(dr, float(g["lr"]), 0, **kw)
    elif kind == "deepcrossing":
        m = models.DeepCrossing(dr, float(g["lr"]), 0, hidden_dim_list=json.loads(str(g["hidden"])),
                                activation_func=str(g["activation"]), normalization=json.loads(str(g["normalization"])),
                                **kw)
    else:
        m = models.DeepFM(dr, float(g["lr"]), 0, hidden_dim_list=json.loads(str(g["hidden"])),
                          normalization=json.loads(str(g["normalization"])), **kw)
    sd = {k[len("init."):]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("init.")}
    assert set(sd) == set(m.state_dict()), set(sd) ^ set(m.state_dict())      # reference key layout
    m.load_state_dict(sd)
    m.train()
    return m, feats


def noise_driven(fname, name):
    """train_deepcrossing_adam_ln.npz: the 1-wide last residual block normalises over one feature, so the gradients
    of its fc and gamma are mathematically zero; the fixture holds torch's rounding noise there (1e-9 .. 1e-6), which
    Adam turns into steps of up to lr.  Nothing downstream depends on them (the block's output is `beta`)."""
    return fname == "train_deepcrossing_adam_ln.npz" and name in ("mlp.2.fc.weight", "mlp.2.fc.bias",
                                                                   "mlp.2.norm_layer.weight")


def compare_state(m, g, prefix, rtol=2e-5, atol=2e-7, only=None):
    have = m.state_dict()
    for k, v in g.items():
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        if only is not None and not only(name):
            continue
        torch.testing.assert_close(have[name].cpu(), torch.from_numpy(np.array(v)), rtol=rtol, atol=atol,
                                   msg=lambda s, name=name: f"{name}: {s}")


def zero_by_construction(g, name):
    """The bias of a Linear in front of a BatchNorm: the batch mean takes it out again, its gradient is exactly zero
    and the fixtures hold torch's rounding noise there (1e-8 .. 1e-7)."""
    return "bn" in str(g["normalization"]) and name.endswith(".fc.bias") and not name.startswith("lr_layer")


GRAD_RTOL = 1e-5          # the tolerance the parity statement names, on the gradients themselves


def compare_grads(have, g, s, fname, skip=lambda n: False, need_all=True):
    """Gradients after the clip against the reference's p.grad after train_step (step{s}.grad.<parameter>):
    |have - want| <= 1e-5 * |want| + 1e-5 * max|want of that parameter|.  The second term is the floor for elements
    that are small sums of large terms (a weight gradient is a sum over the batch of products whose sizes the largest
    element of the same matrix reflects): fp32 sums in another order differ by a few 1e-7 of the TERMS, which no
    tolerance relative to a cancelled result can hold."""
    seen = 0
    for k in g:
        pre = f"step{s}.grad."
        if not k.startswith(pre):
            continue
        n = k[len(pre):]
        if skip(n):
            continue
        if zero_by_construction(g, n):
            assert n in have and float(have[n].abs().max()) <= 1e-6, n
            continue
        if n not in have:
            assert not need_all, f"no gradient for {n}"
            continue
        want = torch.from_numpy(np.array(g[k]))
        floor = GRAD_RTOL * float(want.abs().max())
        torch.testing.assert_close(have[n].detach().cpu(), want, rtol=GRAD_RTOL, atol=floor,
                                   msg=lambda m_, n=n: f"{fname} step {s} grad {n}: {m_}")
        seen += 1
    return seen


@pytest.mark.parametrize("fname", FIXTURES + RESIDUAL_FIXTURES + LONG_FIXTURES + REG_SGD_FIXTURES)
def test_module_path_train_step(fname):
    g = load_golden(fname)
    m, feats = build_model(g)
    for s in range(int(g["steps"])):
        x = golden_batch(g, feats, f"step{s}.x.")
        y = {"label": torch.from_numpy(g[f"step{s}.label"])}
        with torch.no_grad():
            torch.testing.assert_close(m.logits(x).cpu(), torch.from_numpy(g[f"step{s}.logits"]), rtol=1e-5, atol=1e-6)
        loss = m.train_step(x, y)
        torch.testing.assert_close(loss.detach().cpu(), torch.from_numpy(g[f"step{s}.loss"]), rtol=1e-5, atol=1e-7)
        for n, p in m.named_parameters():
            key = f"step{s}.grad.{n}"
            if fname in LONG_FIXTURES:          # weights only
                break
            if noise_driven(fname, n):
                continue
            if p.grad is None:
                assert key not in g
            elif fname == "train_deepcrossing_adam_ln.npz":
                # (torch's LayerNorm over the one feature of the last residual block back-propagates rounding noise
                #  times rstd = 316 instead of an exact zero: the reference's own upstream gradients carry ~2e-4 of it)
                torch.testing.assert_close(p.grad.cpu(), torch.from_numpy(g[key]), rtol=1e-3, atol=5e-6,
                                           msg=lambda s_, n=n: f"{n}: {s_}")
        if fname not in LONG_FIXTURES and fname != "train_deepcrossing_adam_ln.npz":
            compare_grads({n: p.grad for n, p in m.named_parameters() if p.grad is not None}, g, s, fname)
        compare_state(m, g, f"step{s}.after.", only=lambda name: not noise_driven(fname, name),
                      **(dict(rtol=2e-4, atol=2e-6) if fname == "train_deepcrossing_adam_ln.npz" else {}))


@pytest.mark.parametrize("fname", FIXTURES + REG_SGD_FIXTURES)
@pytest.mark.parametrize("graph", [False, True])
def test_fused_engine_dense_mode(fname, graph):
    g = load_golden(fname)
    m, feats = build_model(g)
    eng = m.use_fused_step(mode="dense", use_graph=graph)
    for s in range(int(g["steps"])):
        x = golden_batch(g, feats, f"step{s}.x.")
        y = {"label": torch.from_numpy(g[f"step{s}.label"])}
        idx = m.embedding_layer.index_matrix(m.features_to_divice(x))
        torch.testing.assert_close(eng.logits(idx).cpu(), torch.from_numpy(g[f"step{s}.logits"]), rtol=1e-5, atol=1e-6)
        loss = m.train_step(x, y)
        torch.testing.assert_close(loss.cpu()[0], torch.from_numpy(g[f"step{s}.loss"]), rtol=1e-5, atol=1e-7)
        assert compare_grads(eng.gradients(), g, s, fname) > 0
        compare_state(m, g, f"step{s}.after.")
    m.embedding_layer._tab.layout.check_indices()


def test_touched_rows_step_on_residual_stack_sgd():
    """The reference's train_step on a DeepCrossing stack (ReLU + BatchNorm residual blocks), replayed by the generic
    touched-rows trainer: SGD has no state, so every step equals the reference's dense update."""
    g = load_golden("train_deepcrossing_sgd_bn.npz")
    m, feats = build_model(g)
    m.use_sparse_step()
    for s in range(int(g["steps"])):
        x = golden_batch(g, feats, f"step{s}.x.")
        y = {"label": torch.from_numpy(g[f"step{s}.label"])}
        with torch.no_grad():                   # the fixture ran one extra training-mode forward per step
            torch.testing.assert_close(m.logits(x).cpu(), torch.from_numpy(g[f"step{s}.logits"]), rtol=1e-5, atol=1e-6)
        loss = m.train_step(x, y)
        torch.testing.assert_close(loss.cpu().view(()), torch.from_numpy(g[f"step{s}.loss"]), rtol=1e-5, atol=1e-7)
        compare_state(m, g, f"step{s}.after.")


@pytest.mark.parametrize("fname", ["train_fm_sgd.npz", "train_deepfm_sgd_bn.npz"])
def test_fused_engine_sparse_mode_equals_dense_sgd(fname):
    """SGD has no state: the touched-rows update is exactly the reference's dense update."""
    g = load_golden(fname)
    m, feats = build_model(g)
    eng = m.use_fused_step(mode="sparse")
    for s in range(int(g["steps"])):
        x = golden_batch(g, feats, f"step{s}.x.")
        y = {"label": torch.from_numpy(g[f"step{s}.label"])}
        # the fixture ran one extra training-mode forward per step (BatchNorm running statistics see it)
        eng.logits(m.embedding_layer.index_matrix(m.features_to_divice(x)))
        loss = m.train_step(x, y)
        torch.testing.assert_close(loss.cpu()[0], torch.from_numpy(g[f"step{s}.loss"]), rtol=1e-5, atol=1e-7)
        assert compare_grads(eng.gradients(), g, s, fname) > 0
        compare_state(m, g, f"step{s}.after.")


@pytest.mark.parametrize("fname", REG_SGD_FIXTURES)
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("look_every_step", [True, False])
def test_fused_engine_touched_rows_sgd_with_emb_regular(fname, graph, look_every_step):
    """SGD + emb_regular on the touched-rows step against the reference's traces: the L2 term gives every row of every
    nn.Embedding the gradient lambda * w (BaseModel.py:214-231), so rows off the batch shrink by 1 - lr c lambda per
    step and all of them count in the clipped norm.  The engine keeps the tables as w / G with one scalar G
    (ops.TableShrink) and never visits a row the batch does not touch; state_dict() turns the stored rows back into
    weights.  look_every_step=False leaves the tables scaled for the whole trace and compares at its end."""
    g = load_golden(fname)
    m, feats = build_model(g)
    eng = m.use_fused_step(mode="sparse", use_graph=graph)
    assert eng.shrink is not None and not eng._fuse_small
    steps = int(g["steps"])
    for s in range(steps):
        x = golden_batch(g, feats, f"step{s}.x.")
        y = {"label": torch.from_numpy(g[f"step{s}.label"])}
        idx = m.embedding_layer.index_matrix(m.features_to_divice(x))
        torch.testing.assert_close(eng.logits(idx).cpu(), torch.from_numpy(g[f"step{s}.logits"]), rtol=1e-5, atol=1e-6)
        loss = m.train_step(x, y)
        torch.testing.assert_close(loss.cpu()[0], torch.from_numpy(g[f"step{s}.loss"]), rtol=1e-5, atol=1e-7)
        if look_every_step or s == steps - 1:
            compare_state(m, g, f"step{s}.after.")          # tables the model never looks up included
    if not look_every_step:
        assert abs(float(eng.shrink.state[0]) - 1.0) < 1e-12 and not eng.shrink.dirty      # state_dict() flushed
    m.embedding_layer._tab.layout.check_indices()


def test_touched_rows_emb_regular_survives_load_state_dict():
    """Weights written from outside between two steps (fit() ends with load_weights()): the scale goes back into the
    rows first, and the squared norm of all rows that the clipped norm needs is taken from the new weights."""
    g = load_golden("train_fm_sgd_reg_clip.npz")
    m, feats = build_model(g)
    init = {k: v.clone() for k, v in m.state_dict().items()}
    eng = m.use_fused_step(mode="sparse", use_graph=True)
    for s in range(2):
        m.train_step(golden_batch(g, feats, f"step{s}.x."), {"label": torch.from_numpy(g[f"step{s}.label"])})
    assert eng.shrink.dirty
    m.load_state_dict(init)
    for s in range(int(g["steps"])):
        x = golden_batch(g, feats, f"step{s}.x.")
        loss = m.train_step(x, {"label": torch.from_numpy(g[f"step{s}.label"])})
        torch.testing.assert_close(loss.cpu()[0], torch.from_numpy(g[f"step{s}.loss"]), rtol=1e-5, atol=1e-7)
    compare_state(m, g, f"step{int(g['steps']) - 1}.after.")


def test_touched_rows_emb_regular_needs_sgd_on_one_gpu():
    from recommendsystem_amd._lib import RSError
    g = load_golden("train_lr_adam_reg.npz")
    m, _ = build_model(g)
    with pytest.raises(RSError):
        m.use_fused_step(mode="sparse")             # Adam: every row moves by its own amount in every step


@pytest.mark.parametrize("fname", ["train_deepfm_adam.npz", "train_deepfm_adam_long.npz", "train_deepfm_adamw_long.npz"])
@pytest.mark.parametrize("graph", [False, True])
def test_fused_engine_sparse_mode_exact_adam(fname, graph):
    """Touched-rows Adam / AdamW with the skipped zero-gradient steps of a row replayed before it is next read: every
    step of the reference's traces (up to 12 steps; rows stay absent for several steps in a row while their moments
    still move them in the reference), logits included.  state_dict() brings all rows up to date before comparing."""
    g = load_golden(fname)
    m, feats = build_model(g)
    eng = m.use_fused_step(mode="sparse", use_graph=graph)
    assert eng.exact
    for s in range(int(g["steps"])):
        x = golden_batch(g, feats, f"step{s}.x.")
        y = {"label": torch.from_numpy(g[f"step{s}.label"])}
        idx = m.embedding_layer.index_matrix(m.features_to_divice(x))
        torch.testing.assert_close(eng.logits(idx).cpu(), torch.from_numpy(g[f"step{s}.logits"]), rtol=1e-5, atol=1e-6)
        loss = m.train_step(x, y)
        torch.testing.assert_close(loss.cpu()[0], torch.from_numpy(g[f"step{s}.loss"]), rtol=1e-5, atol=1e-7)
        if fname not in LONG_FIXTURES:                      # (the long traces hold weights only)
            assert compare_grads(eng.gradients(), g, s, fname) > 0
        if s % 3 == 2 or s == int(g["steps"]) - 1:          # in between, rows stay behind for several steps
            compare_state(m, g, f"step{s}.after.", only=lambda n: "embedding_layer.embedded_to_1_dim" not in n
                          and "lr_layer.scalar.embedded_features" not in n)
    m.embedding_layer._tab.layout.check_indices()


def test_module_forward_reads_caught_up_rows():
    """A reader outside the engine (the model's own forward) sees the rows the engine left behind brought up to date."""
    g = load_golden("train_deepfm_adam_long.npz")
    m, feats = build_model(g)
    m.use_fused_step(mode="sparse")
    for s in range(int(g["steps"])):
        x = golden_batch(g, feats, f"step{s}.x.")
        with torch.no_grad():
            torch.testing.assert_close(m.logits(x).cpu(), torch.from_numpy(g[f"step{s}.logits"]), rtol=1e-5, atol=1e-6)
        m.train_step(x, {"label": torch.from_numpy(g[f"step{s}.label"])})


def test_fused_engine_sparse_adam_first_step_and_lazy_rows():
    """exact_adam=False: step 1 of Adam from zero state is identical in lazy and dense form; afterwards rows absent
    from a batch keep their weights in lazy mode while every other parameter still tracks the reference."""
    g = load_golden("train_deepfm_adam.npz")
    m, feats = build_model(g)
    m.use_fused_step(mode="sparse", exact_adam=False)
    x = golden_batch(g, feats, "step0.x.")
    y = {"label": torch.from_numpy(g["step0.label"])}
    m.train_step(x, y)
    compare_state(m, g, "step0.after.")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    x1 = golden_batch(g, feats, "step1.x.")
    m.train_step(x1, {"label": torch.from_numpy(g["step1.label"])})
    compare_state(m, g, "step1.after.", only=lambda n: "embedded" not in n, rtol=1e-4, atol=1e-6)
    fm, _ = golden_feature_map(g)
    for c in ["C1", "C5"]:
        key = f"embedding_layer.embedded_features.{c}.weight"
        touched = set(x1[c].tolist())
        after = m.state_dict()[key]
        for r in range(fm[c]["vocab_size"]):
            if r not in touched:
                assert torch.equal(after[r], before[key][r])


def test_tables_keep_their_storage_across_load_weights(tmp_path):
    """fit() ends with load_weights(), which calls model.to(device) and load_state_dict(): the stacked tables must stay
    the tensors the engine (and its captured graphs) hold, and training must go on moving the model's own tables."""
    g = load_golden("train_deepfm_adam.npz")
    m, feats = build_model(g)
    eng = m.use_fused_step(mode="sparse", use_graph=True)
    x = golden_batch(g, feats, "step0.x.")
    y = {"label": torch.from_numpy(g["step0.label"])}
    m.train_step(x, y)
    ptr0 = m.embedding_layer._tab.stacked.data_ptr()
    path = str(tmp_path / "w.model")
    torch.save(m.state_dict(), path)
    m.load_weights(path)
    assert m.embedding_layer._tab.stacked.data_ptr() == ptr0
    for c in m.embedding_layer.lookup_columns:            # per-column parameters are still views of the stacked table
        w = m.embedding_layer.embedded_features[c].weight
        assert ptr0 <= w.data_ptr() < ptr0 + m.embedding_layer._tab.stacked.numel() * 4
    before = m.embedding_layer._tab.stacked.clone()
    m.train_step(x, y)
    assert (m.embedding_layer._tab.stacked != before).any()
    assert eng.step_count.item() == 2


def test_out_of_range_index_raises_like_nn_embedding():
    g = load_golden("train_fm_sgd.npz")
    m, feats = build_model(g)
    eng = m.use_fused_step(mode="sparse")
    x = golden_batch(g, feats, "step0.x.")
    x = {c: v.clone() for c, v in x.items()}
    x["C2"][3] = 10 ** 6
    m.train_step(x, {"label": torch.from_numpy(g["step0.label"])})
    with pytest.raises(IndexError):
        eng.check()


def test_out_of_range_index_is_reported_by_the_periodic_poll():
    """Nobody calls check(): the flag a kernel raised is found by the step's own periodic, non-synchronising look at
    the flags (every check_every steps; the snapshot requested at one look is read at the next), within three looks."""
    g = load_golden("train_fm_sgd.npz")
    m, feats = build_model(g)
    eng = m.use_fused_step(mode="sparse")
    eng.check_every = 2
    y = {"label": torch.from_numpy(g["step0.label"])}
    good = golden_batch(g, feats, "step0.x.")
    for _ in range(5):                          # a clean run raises nothing
        m.train_step(good, y)
    bad = {c: v.clone() for c, v in good.items()}
    bad["C2"][3] = 10 ** 6
    m.train_step(bad, y)
    with pytest.raises(IndexError):
        for _ in range(3 * eng.check_every + 1):
            m.train_step(good, y)


def _deepfm_pair_states(betas, opt, steps=14, start_step=0):
    """The same model stepped by the dense sweep and by the touched-rows step with replayed skips; (dense, exact).
    start_step: the optimizers' step counters start there (every row up to date at that step)."""
    from recommendsystem_amd import data, models
    from recommendsystem_amd.base_model import random_all
    vocab = [7, 40, 300, 2500, 20000, 3]
    out = []
    for mode in ("dense", "sparse"):
        random_all(5)
        rec = data.criteo_x1_recorder(embedding_dim=8, batch_size=64, vocab_sizes=vocab)
        m = models.DeepFM(rec, 1e-2, 0, hidden_dim_list=[32, 16], normalization=None, optimizer_type=opt)
        m.train()
        eng = m.use_fused_step(mode=mode, betas=betas)
        if start_step:
            eng.step_count.fill_(start_step)
            for t in eng.tables:
                if t.get("rs") is not None:
                    t["rs"].last.fill_(start_step)
        idxs, labs = data.synthetic_batches(vocab, 64, steps, "cuda", seed=9)
        order = [0, 1, 2, 0, 3, 4, 5, 1, 6, 7, 8, 2, 9, 0]
        for s in range(steps):
            eng.step_tensors(idxs[order[s]], labs[order[s]])
        eng.check()
        eng.flush()
        torch.cuda.synchronize()
        out.append({k: v.detach().clone() for k, v in m.state_dict().items()})
    return out


@pytest.mark.parametrize("opt", ["Adam", "AdamW"])
@pytest.mark.parametrize("betas", [(0.9, 0.999), (0.8, 0.99), (0.95, 0.95), (0.0, 0.999)])
def test_exact_adam_other_betas_equals_dense_sweep(betas, opt):
    """The replay's cut-off follows the betas in use (terms decay like beta1 / sqrt(beta2)): touched-rows Adam with the
    skipped steps replayed equals the dense sweep over all rows for coefficients other than torch's defaults too."""
    dense, exact = _deepfm_pair_states(betas, opt)
    assert dense.keys() == exact.keys()
    # atol: 1e-3 of one step (lr = 1e-2; a replay that went wrong is off by whole steps).  A rare row's first gradient can be as small as eps (1e-7 .. 1e-8), where
    # Adam's step lr * m / (sqrt(v) + eps) is no longer scale-free in g, and such a gradient carries the 1e-4 relative
    # noise of everything upstream (the two runs differ by rounding in other rows from the first replay on).
    for k in dense:
        torch.testing.assert_close(exact[k], dense[k], rtol=2e-5, atol=1e-5, msg=lambda s, k=k: f"{k}: {s}")


@pytest.mark.parametrize("betas", [(0.9, 0.999), (0.9, 0.9999)])
def test_exact_adam_past_the_end_of_the_bias_correction_table(betas):
    """The replay reads 1/(1 - beta1^k) and 1/sqrt(1 - beta2^k) from a table of 65536 steps.  Past its end both are
    exactly 1 in float for torch's default betas (the replay then keeps reading the last entry); with beta2 = 0.9999 they
    are not yet, and the replay computes them in double precision as the table's builder does.  Fourteen steps across
    step 65536 against the dense sweep."""
    dense, exact = _deepfm_pair_states(betas, "Adam", start_step=65530)
    # atol = 1e-3 of one step, as above.  Without a bias correction to speak of (1 - beta2^65530 is 0.9986 resp. 1) a first
    # step is lr * (1 - beta1) g / (sqrt(1 - beta2) |g| + eps): 0.032 for beta2 = 0.999 as long as |g| >> 3e-7, and 0.1
    # for beta2 = 0.9999 as long as |g| >> 1e-6 -- ten times the step and three times the range of gradients in which
    # the step follows g's rounding noise
    atol = 1e-5 if betas[1] < 0.9995 else 1e-4
    for k in dense:
        torch.testing.assert_close(exact[k], dense[k], rtol=2e-5, atol=atol, msg=lambda s, k=k: f"{k}: {s}")


def test_exact_adam_refuses_betas_whose_drift_does_not_decay():
    from recommendsystem_amd import data, models
    from recommendsystem_amd._lib import RSError
    rec = data.criteo_x1_recorder(embedding_dim=8, batch_size=16, vocab_sizes=[5, 9])
    m = models.DeepFM(rec, 1e-2, 0, hidden_dim_list=[8], normalization=None, optimizer_type="Adam")
    with pytest.raises(RSError, match="sqrt"):
        m.use_fused_step(mode="sparse", betas=(0.99, 0.9))
    m.use_fused_step(mode="sparse", betas=(0.99, 0.9), exact_adam=False)          # the lazy form has nothing to replay
