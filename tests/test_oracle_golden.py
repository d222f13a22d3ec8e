"""Pins the CPU oracle (oracle/ctr_oracle.py) to vectors captured from the reference's Python.

The fixtures under tests/golden/ were written by oracle/gen_golden.py, which imports the
reference; nothing here reads /root/reference.
"""
import json

import numpy as np
import pytest
import torch

from conftest import (golden_batch, golden_feature_map, golden_state, load_golden,
                      oracle_model_from_golden)
from oracle import ctr_oracle as O

torch.set_num_threads(1)


def T(a):
    return torch.from_numpy(np.array(a))


def test_embedding_lookup_and_lr_bit_exact():
    g = load_golden("layers.npz")
    fm, feats = golden_feature_map(g)
    cols = O.categorical_columns(fm, feats)
    assert cols == ["C1", "C2", "C3", "C4", "C5"]
    x = golden_batch(g, feats)
    tables = {c: T(g[f"table.{c}"]) for c in cols}
    tables1 = {c: T(g[f"table1.{c}"]) for c in cols}
    out = O.embedding_lookup(tables, cols, x)
    assert torch.equal(out, T(g["emb_out"]))
    lr = O.lr_forward(tables1, T(g["lr_bias"]), cols, x)
    assert torch.equal(lr, T(g["lr_out"]))
    for c in cols:                      # padding row is zero, as the reference initialises it
        assert torch.count_nonzero(tables[c][0]) == 0


def test_embedding_dense_grad_row_sets():
    g = load_golden("layers.npz")
    fm, feats = golden_feature_map(g)
    cols = O.categorical_columns(fm, feats)
    x = golden_batch(g, feats)
    go = T(g["emb_grad_out"])
    for f, c in enumerate(cols):
        ref = T(g[f"table_grad.{c}"])
        mine = torch.zeros_like(ref)
        idx = x[c].long()
        for b in range(idx.numel()):          # ascending sample order, padding row skipped
            if idx[b] != 0:
                mine[idx[b]] += go[b, f]
        assert torch.equal(mine, ref)
        touched = set(int(i) for i in idx if i != 0)
        assert set(torch.nonzero(ref.abs().sum(1)).flatten().tolist()) <= touched


@pytest.mark.parametrize("name", json.loads(str(load_golden("mlp_blocks.npz")["names"])))
def test_mlp_blocks(name):
    g = load_golden("mlp_blocks.npz")
    parts = name.split("_")
    act, norm = parts[1], (None if parts[2] == "None" else parts[2])
    p = {k[len(name) + 3:]: T(v) for k, v in g.items() if k.startswith(name + ".p.")}
    x = T(g[name + ".x"]).requires_grad_(True)
    params = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    if name.startswith("mlp_"):
        y = O.mlp_block(x, params["fc.weight"], params["fc.bias"], act, norm,
                        params.get("norm_layer.weight"), params.get("norm_layer.bias"),
                        params.get("activation_func.weight"))
    else:
        y = O.residual_mlp_block(x, params["fc.weight"], params["fc.bias"], act, norm,
                                 params["norm_layer.weight"], params["norm_layer.bias"],
                                 params.get("downsample.weight"), params.get("downsample.bias"))
    torch.testing.assert_close(y, T(g[name + ".y"]), rtol=1e-6, atol=1e-7)
    (y * T(g[name + ".g"])).sum().backward()
    torch.testing.assert_close(x.grad, T(g[name + ".dx"]), rtol=1e-5, atol=1e-6)
    for k, v in params.items():
        torch.testing.assert_close(v.grad, T(g[f"{name}.dp.{k}"]), rtol=1e-5, atol=1e-6)


def test_interactions():
    g = load_golden("interactions.npz")
    B, F, d = int(g["B"]), int(g["F"]), int(g["d"])
    emb, lr, xi = T(g["emb"]), T(g["lr"]), T(g["xi"])
    torch.testing.assert_close(lr + O.fm_second_order(emb), T(g["fm_out"]), rtol=1e-6, atol=1e-6)
    # FM second order equals the explicit pair sum
    pair = sum((emb[:, i] * emb[:, j]).sum(1, keepdim=True) for i in range(F) for j in range(i + 1, F))
    torch.testing.assert_close(O.fm_second_order(emb), pair, rtol=1e-5, atol=1e-5)
    # DeepFM
    h = O.mlp_block(emb, T(g["deepfm.deep.0.fc.weight"]), T(g["deepfm.deep.0.fc.bias"]))
    h = O.mlp_block(h, T(g["deepfm.deep.1.fc.weight"]), T(g["deepfm.deep.1.fc.bias"]))
    torch.testing.assert_close(lr + O.fm_second_order(emb) + h, T(g["deepfm_out"]), rtol=1e-6, atol=1e-6)
    # FFM
    tabs = [[T(g[f"ffm.table.{i}.{j}"]) for j in range(F)] for i in range(F)]
    torch.testing.assert_close(O.ffm_interaction(tabs, xi), T(g["ffm_out"]), rtol=1e-6, atol=1e-6)
    # PNN
    flat = emb.reshape(B, -1)
    assert torch.equal(O.pnn_inner(flat, F, d), T(g["pnn_inner"]))
    assert torch.equal(O.pnn_outer(flat, F, d), T(g["pnn_outer"]))
    assert torch.equal(torch.cat([O.pnn_inner(flat, F, d), O.pnn_outer(flat, F, d)], 1), T(g["pnn_both"]))
    # NFM
    bi = O.nfm_bi_interaction_reference(flat, T(g["nfm_v"]))
    torch.testing.assert_close(bi, T(g["nfm_bi"]), rtol=1e-6, atol=1e-6)
    first = xi.float() @ T(g["nfm_lin_w"]).t() + T(g["nfm_lin_b"])
    torch.testing.assert_close(torch.sigmoid(first + bi.sum(1, keepdim=True)), T(g["nfm_out"]), rtol=1e-6, atol=1e-6)
    # AFM
    e3 = T(g["afm_emb"])
    assert torch.equal(O.afm_pairwise(e3), T(g["afm_pairwise"]))
    term = O.afm_forward(e3, T(g["afm.att_w"]), T(g["afm.att_b"]), T(g["afm.att_h"]), T(g["afm.p_w"]), T(g["afm.p_b"]))
    torch.testing.assert_close(torch.sigmoid(lr + term), T(g["afm_out"]), rtol=1e-6, atol=1e-6)
    # DCN
    out = O.dcn_cross(flat, [T(g[f"dcn.w.{i}"]) for i in range(3)], [T(g[f"dcn.b.{i}"]) for i in range(3)])
    torch.testing.assert_close(out, T(g["dcn_out"]), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("fname", ["train_lr_adam_reg.npz", "train_fm_sgd.npz", "train_deepfm_adam.npz",
                                   "train_deepfm_sgd_bn.npz", "train_deepfm_adamw_reg.npz",
                                   "train_deepcrossing_sgd_bn.npz", "train_deepcrossing_adam_ln.npz",
                                   "train_deepfm_adam_long.npz", "train_deepfm_adamw_long.npz",
                                   "train_lr_sgd_reg.npz", "train_deepfm_sgd_reg.npz", "train_fm_sgd_reg_clip.npz"])
def test_train_steps(fname):
    g = load_golden(fname)
    fm, feats = golden_feature_map(g)
    m = oracle_model_from_golden(g)
    m.train()
    reg = float(g["regular"])
    reg = None if reg < 0 else reg
    opt = O.make_optimizer(str(g["optimizer"]), m.parameters(), float(g["lr"]))
    for s in range(int(g["steps"])):
        x = golden_batch(g, feats, f"step{s}.x.")
        y = T(g[f"step{s}.label"])
        with torch.no_grad():
            torch.testing.assert_close(m(x), T(g[f"step{s}.logits"]), rtol=1e-5, atol=1e-6)
        loss, _ = O.train_step(m, opt, x, y, emb_regular=reg, net_regular=reg)
        torch.testing.assert_close(loss, T(g[f"step{s}.loss"]), rtol=1e-6, atol=1e-7)
        want = golden_state(g, f"step{s}.after.")
        have = m.state_dict()
        for k, v in have.items():
            torch.testing.assert_close(v, want[k], rtol=1e-5, atol=1e-7, msg=lambda s_, k=k: f"{k}: {s_}")
        grads = golden_state(g, f"step{s}.grad.")
        if not grads:                       # the long traces keep weights only
            continue
        for n, p in m.named_parameters():
            if p.grad is None:
                assert n not in grads
            else:
                torch.testing.assert_close(p.grad, grads[n], rtol=1e-5, atol=1e-8, msg=lambda s_, n=n: f"{n}: {s_}")


def test_state_dict_layout_fixture():
    import os
    from conftest import GOLDEN
    layout = json.load(open(os.path.join(GOLDEN, "lr_state_dict_layout.json")))
    assert layout["lr_layer.bias"] == [1]
    assert layout["embedding_layer.embedded_features.userid.weight"] == [26, 1]
    assert layout["lr_layer.scalar.embedded_to_1_dim_features.adgroup_id.weight"] == [96, 1]


def test_auc_matches_sklearn():
    from sklearn.metrics import roc_auc_score
    g = torch.Generator().manual_seed(3)
    y = (torch.rand(500, generator=g) < 0.3).long()
    s = torch.round(torch.rand(500, generator=g) * 50) / 50      # many ties
    assert abs(O.auc(y, s) - roc_auc_score(y.numpy(), s.numpy())) < 1e-12
