"""Generate tests/golden/*.npz by running the reference's own Python on CPU.

TEST INFRASTRUCTURE.  Run in the build container only (the reference lives at
/root/reference there and never travels):

    python oracle/gen_golden.py

What it does: imports the reference's live modules (model.utils.Layers,
model.utils.BaseModel), aliases the stale package name ``models`` to ``model``
so the nine model files import, calls their methods on stub ``self`` objects,
and saves inputs/outputs as small arrays.  Two harness-side accommodations of
reference defects:

* ``Embedding_layer.forward(scalar_only=True)`` raises KeyboardInterrupt
  (model/utils/Layers.py:154-157), so LR vectors are produced from the
  per-column ``embedded_to_1_dim_features[col]`` modules directly followed by
  the unreachable line's intent (``stack(dim=1)``) and Layers.py:109.
* ``DCN.cross_b`` is created (D, 1); the stub passes (1, D) so the line
  model/DCN/DCN.py:51 broadcasts as intended.
"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

sys.path.insert(0, REF)
import model  # noqa: E402  (namespace package of the reference)
import model.utils  # noqa: E402
import model.utils.Layers as RL  # noqa: E402
import model.utils.BaseModel as RB  # noqa: E402
from model.utils.utils import random_all  # noqa: E402

sys.modules["models"] = model
sys.modules["models.utils"] = model.utils
sys.modules["models.utils.Layers"] = RL
sys.modules["models.utils.BaseModel"] = RB
from model.FM.FM import FM  # noqa: E402
from model.DeepFM.DeepFM import DeepFM  # noqa: E402
from model.FFM.FFM import FFM  # noqa: E402
from model.PNN.PNN import PNN  # noqa: E402
from model.NFM.NFM import NFM  # noqa: E402
from model.AFM.AFM import AFM  # noqa: E402
from model.DCN.DCN import DCN  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy().copy()     # copy: parameters are updated in place later


def stub_module():
    m = object.__new__(nn.Module)
    nn.Module.__init__(m)
    return m


def recorder(vocabs, n_num, d, bs=8):
    fm = {}
    cat = [f"C{i + 1}" for i in range(len(vocabs))]
    num = [f"I{i + 1}" for i in range(n_num)]
    for c, v in zip(cat, vocabs):
        fm[c] = {"type": "categorical", "vocab_size": int(v)}
    for c in num:
        fm[c] = {"type": "numerical"}
    feats = cat + num
    return SimpleNamespace(embedding_dim=d, feature_map=fm, features=feats, feature_num=len(feats),
                           input_dim=len(cat) * d, batch_size=bs, categorical_col=cat, numerical_col=num)


def batch(dr, bs, gen, force_edges=True):
    x = {}
    for c in dr.features:
        info = dr.feature_map[c]
        if info["type"] == "categorical":
            v = info["vocab_size"]
            t = torch.randint(1, v, (bs,), generator=gen)
            if force_edges and bs >= 4:
                t[0] = 0          # PAD row
                t[1] = v - 1      # OOV row (last)
                t[2] = t[3]       # a duplicate inside the batch
            x[c] = t
        else:
            x[c] = torch.rand(bs, generator=gen)
    return x


def lr_from_modules(emb_layer, bias, x):
    """Layers.py:150-151 per column, then the intent of :158 and :109."""
    cols = [c for c in emb_layer.dataRecorder.features
            if emb_layer.feature_map[c]["type"] == "categorical"]
    rows = [emb_layer.embedded_to_1_dim_features[c](x[c].long()) for c in cols]
    return torch.stack(rows, dim=1).sum(dim=1) + bias


# ---------------------------------------------------------------- layers
def gen_layers():
    random_all(2021)
    gen = torch.Generator().manual_seed(7)
    dr = recorder([11, 7, 50, 3, 200], 2, 4)
    emb = RL.Embedding_layer(dr)
    x = batch(dr, 8, gen)
    out = emb(x)
    bias = torch.tensor([0.03125])
    lr = lr_from_modules(emb, bias, x)
    d = {"feature_map": json.dumps(dr.feature_map), "features": json.dumps(dr.features),
         "emb_out": npy(out), "lr_out": npy(lr), "lr_bias": npy(bias), "dim": 4}
    for c in dr.categorical_col:
        d[f"table.{c}"] = npy(emb.embedded_features[c].weight)
        d[f"table1.{c}"] = npy(emb.embedded_to_1_dim_features[c].weight)
    for c in dr.features:
        d[f"x.{c}"] = npy(x[c])
    # gradient of sum(out * g) w.r.t. every table (dense, padding row 0 stays zero)
    g = torch.randn(out.shape, generator=gen)
    (out * g).sum().backward()
    d["emb_grad_out"] = npy(g)
    for c in dr.categorical_col:
        d[f"table_grad.{c}"] = npy(emb.embedded_features[c].weight.grad)
    np.savez(os.path.join(OUT, "layers.npz"), **d)


# ---------------------------------------------------------------- MLP blocks
def gen_mlp():
    random_all(2021)
    gen = torch.Generator().manual_seed(11)
    d = {}
    cases = [("relu", None), ("relu", "bn"), ("tanh", "ln"), ("leakyrelu", "bn"), ("prelu", None),
             ("tanh", None), ("leakyrelu", "ln")]
    names = []
    for act, norm in cases:
        name = f"mlp_{act}_{norm}"
        names.append(name)
        blk = RL.MLPBlock(12, 5, act, norm)
        blk.train()
        with torch.no_grad():
            if norm:
                blk.norm_layer.weight.copy_(torch.rand(5, generator=gen) + 0.5)
                blk.norm_layer.bias.copy_(torch.randn(5, generator=gen) * 0.1)
        x = torch.randn(8, 3, 4, generator=gen, requires_grad=True)   # any leading shape, flattened inside
        y = blk(x)
        g = torch.randn(y.shape, generator=gen)
        (y * g).sum().backward()
        d[f"{name}.x"] = npy(x); d[f"{name}.y"] = npy(y); d[f"{name}.g"] = npy(g)
        d[f"{name}.dx"] = npy(x.grad)
        for pn, p in blk.named_parameters():
            d[f"{name}.p.{pn}"] = npy(p)
            d[f"{name}.dp.{pn}"] = npy(p.grad)
        if norm == "bn":
            d[f"{name}.running_mean"] = npy(blk.norm_layer.running_mean)
            d[f"{name}.running_var"] = npy(blk.norm_layer.running_var)
            blk.eval()
            d[f"{name}.y_eval"] = npy(blk(x))
    rcases = [("relu", "bn", 12, 5), ("tanh", "ln", 6, 6), ("leakyrelu", "bn", 6, 6)]
    for act, norm, i, o in rcases:
        name = f"res_{act}_{norm}_{i}_{o}"
        names.append(name)
        blk = RL.ResidualMLPBlock(i, o, act, norm)
        blk.train()
        with torch.no_grad():
            blk.norm_layer.weight.copy_(torch.rand(o, generator=gen) + 0.5)
            blk.norm_layer.bias.copy_(torch.randn(o, generator=gen) * 0.1)
        x = torch.randn(8, i, generator=gen, requires_grad=True)
        y = blk(x)
        g = torch.randn(y.shape, generator=gen)
        (y * g).sum().backward()
        d[f"{name}.x"] = npy(x); d[f"{name}.y"] = npy(y); d[f"{name}.g"] = npy(g)
        d[f"{name}.dx"] = npy(x.grad)
        for pn, p in blk.named_parameters():
            d[f"{name}.p.{pn}"] = npy(p)
            d[f"{name}.dp.{pn}"] = npy(p.grad)
    d["names"] = json.dumps(names)
    np.savez(os.path.join(OUT, "mlp_blocks.npz"), **d)


# ---------------------------------------------------------------- interactions
def gen_interactions():
    random_all(2021)
    gen = torch.Generator().manual_seed(13)
    B, Fn, dim = 6, 5, 4
    d = {"B": B, "F": Fn, "d": dim}
    emb = torch.randn(B, Fn, dim, generator=gen)
    lr = torch.randn(B, 1, generator=gen)
    xi = torch.randint(0, 9, (B, Fn), generator=gen)
    d["emb"] = npy(emb); d["lr"] = npy(lr); d["xi"] = npy(xi)

    # FM.forward / DeepFM.forward on a stub self (FM.py:19-33, DeepFM.py:42-55)
    s = stub_module()
    s.device = "cpu"
    s.linear = lambda x: lr
    s.embedding_layer = lambda x: emb
    d["fm_out"] = npy(FM.forward(s, xi))
    deep = nn.Sequential(RL.MLPBlock(Fn * dim, 8, "ReLU", None), RL.MLPBlock(8, 1, "ReLU", None))
    s.deep = deep
    d["deepfm_out"] = npy(DeepFM.forward(s, xi))
    for pn, p in deep.named_parameters():
        d[f"deepfm.deep.{pn}"] = npy(p)

    # FFM.ffm_interaction with independent F x F tables (FFM.py:43-62)
    s = stub_module()
    s.device = "cpu"
    s.feature_num = Fn
    s.feature_name_list = [f"f{i}" for i in range(Fn)]
    s.field_index = torch.arange(Fn)
    s.embedding_dict = {}
    for i in range(Fn):
        tabs = []
        for j in range(Fn):
            e = nn.Embedding(9, dim)
            d[f"ffm.table.{i}.{j}"] = npy(e.weight)
            tabs.append(e)
        s.embedding_dict[f"f{i}"] = tabs
    d["ffm_out"] = npy(FFM.ffm_interaction(s, xi))

    # PNN (PNN.py:61-103)
    s = stub_module()
    s.feature_num = Fn
    s.dataRecorder = SimpleNamespace(embedding_dim=dim)
    flat = emb.view(B, -1)
    d["pnn_inner"] = npy(PNN.inner_product(s, flat))
    d["pnn_outer"] = npy(PNN.outer_product(s, flat))
    s.inner_product = lambda t: PNN.inner_product(s, t)
    s.outer_product = lambda t: PNN.outer_product(s, t)
    d["pnn_both"] = npy(PNN.both_product(s, flat))

    # NFM.forward (NFM.py:37-49) with flattened embeddings (2-D, as `emb_out @ v` needs)
    s = stub_module()
    s.device = "cpu"
    k = 3
    s.v = torch.randn(Fn * dim, k, generator=gen)
    s.linear = nn.Linear(Fn, 1)
    s.embedding_layer = lambda x: flat
    captured = {}

    class Tap(nn.Module):
        def forward(self, z):
            captured["bi"] = z
            return z.sum(dim=1, keepdim=True)
    s.deep = Tap()
    d["nfm_out"] = npy(NFM.forward(s, xi))
    d["nfm_bi"] = npy(captured["bi"])
    d["nfm_v"] = npy(s.v)
    d["nfm_lin_w"] = npy(s.linear.weight); d["nfm_lin_b"] = npy(s.linear.bias)

    # AFM (AFM.py:19-56)
    s = stub_module()
    s.device = "cpu"
    s.emb_dim = dim
    s.feature_num = Fn
    A = 3
    s.attention = nn.Sequential(nn.Linear(dim, A), nn.ReLU(), nn.Linear(A, 1, bias=False), nn.Softmax(dim=1))
    s.p = nn.Linear(dim, 1)
    s.linear = lambda x: lr
    s.embedding_layer = lambda x: emb
    s.pair_wise_interaction_layer = lambda e: AFM.pair_wise_interaction_layer(s, e)
    emb_big = emb * 3.0   # larger activations so the softmax is not flat
    s.embedding_layer = lambda x: emb_big
    d["afm_emb"] = npy(emb_big)
    d["afm_pairwise"] = npy(AFM.pair_wise_interaction_layer(s, emb_big))
    d["afm_out"] = npy(AFM.forward(s, xi))       # sigmoid(lr + afm term)
    d["afm.att_w"] = npy(s.attention[0].weight); d["afm.att_b"] = npy(s.attention[0].bias)
    d["afm.att_h"] = npy(s.attention[2].weight)
    d["afm.p_w"] = npy(s.p.weight); d["afm.p_b"] = npy(s.p.bias)

    # DCN.crossing_network (DCN.py:48-52), bias passed as (1, D)
    s = stub_module()
    D = Fn * dim
    s.cross_layers_num = 3
    s.cross_w = [torch.randn(D, 1, generator=gen) * 0.3 for _ in range(3)]
    s.cross_b = [torch.randn(1, D, generator=gen) * 0.3 for _ in range(3)]
    d["dcn_out"] = npy(DCN.crossing_network(s, flat))
    for i in range(3):
        d[f"dcn.w.{i}"] = npy(s.cross_w[i]); d[f"dcn.b.{i}"] = npy(s.cross_b[i])
    np.savez(os.path.join(OUT, "interactions.npz"), **d)


# ---------------------------------------------------------------- training step
class RefModel(RB.BaseModel):
    """A model written against the reference's LIVE API (the way model/LR/LR.py is):
    reference BaseModel + reference Layers; forward composes them as the stale model
    files specify (LR.py:29-33, FM.py:19-33, DeepFM.py:42-55) and ends in sigmoid so that
    BaseModel.train_step's F.binary_cross_entropy applies (LR.py:32)."""

    def __init__(self, kind, dr, lr, optimizer_type, hidden=(), activation="ReLU", normalization=None,
                 regular=None):
        super().__init__(model_name=kind, dataRecorder=dr, learning_rate=lr, optimizer_type=optimizer_type,
                         gpu=-1, emb_regular=regular, net_regular=regular)
        self.kind = kind
        if kind != "deepcrossing":
            self.lr_layer = RL.LinearRegression_layer(dr)
        if kind == "deepcrossing":          # model/DeepCrossing/DeepCrossing.py:15-26
            dims = [dr.input_dim] + list(hidden) + [1]
            self.mlp = nn.Sequential(*[RL.ResidualMLPBlock(dims[i], dims[i + 1], activation, normalization)
                                       for i in range(len(dims) - 1)])
        if kind == "deepfm":
            dims = [dr.input_dim] + list(hidden)
            layers = [RL.MLPBlock(dims[i], dims[i + 1], activation, normalization) for i in range(len(dims) - 1)]
            layers.append(RL.MLPBlock(dims[-1], 1, activation, normalization))
            self.deep = nn.Sequential(*layers)
        self.initialize_criterion()
        self.initialize_optimizer()
        self.reset_parameters()
        self.model_to_device()

    def logits(self, x):
        x = self.features_to_divice(x)
        if self.kind == "deepcrossing":     # model/DeepCrossing/DeepCrossing.py:32-36 (before its sigmoid)
            return self.mlp(self.embedding_layer(x))
        lr = lr_from_modules(self.lr_layer.scalar, self.lr_layer.bias, x)
        if self.kind == "lr":
            return lr
        emb_out = self.embedding_layer(x)
        square_of_sum = torch.pow(torch.sum(emb_out, dim=1), 2)
        sum_of_square = torch.sum(torch.pow(emb_out, 2), dim=1)
        second = 0.5 * torch.sum((square_of_sum - sum_of_square), dim=1, keepdim=True)
        out = lr + second
        if self.kind == "deepfm":
            out = out + self.deep(emb_out)
        return out

    def forward(self, x):
        return torch.sigmoid(self.logits(x))


def gen_train(kind, optimizer_type, regular, fname, hidden=(), normalization=None, scale_tables=1.0,
              lr=1e-3, steps=3, bs=16, vocabs=(11, 7, 50, 3, 200), dim=4, activation="ReLU", slim=False):
    random_all(2021)
    gen = torch.Generator().manual_seed(17)
    dr = recorder(list(vocabs), 2, dim, bs)
    m = RefModel(kind, dr, lr, optimizer_type, hidden=hidden, activation=activation, normalization=normalization,
                 regular=regular)
    if scale_tables != 1.0:      # larger embeddings so second-order terms matter in the test
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "embedded" in n:
                    p.mul_(scale_tables)
    m.train()
    d = {"feature_map": json.dumps(dr.feature_map), "features": json.dumps(dr.features), "dim": dim,
         "kind": kind, "optimizer": optimizer_type, "lr": lr, "regular": -1.0 if regular is None else regular,
         "hidden": json.dumps(list(hidden)), "normalization": json.dumps(normalization), "steps": steps,
         "activation": activation}
    for n, p in m.state_dict().items():
        d[f"init.{n}"] = npy(p)
    for s in range(steps):
        x = batch(dr, bs, gen, force_edges=(s == 0))
        y = {"label": (torch.rand(bs, generator=gen) < 0.3).long()}
        inputs = dict(x)
        inputs["label"] = y["label"]            # the dataset's feature dict carries every column (PreProcess.py:49)
        with torch.no_grad():
            d[f"step{s}.logits"] = npy(m.logits(inputs))
        loss = m.train_step(inputs, y)          # BaseModel.py:141-150, unchanged
        d[f"step{s}.loss"] = npy(loss)
        for c in dr.features:
            d[f"step{s}.x.{c}"] = npy(x[c])
        d[f"step{s}.label"] = npy(y["label"])
        for n, p in m.named_parameters():
            if p.grad is not None and not slim:      # unused tables keep grad None and are not stepped
                d[f"step{s}.grad.{n}"] = npy(p.grad)     # after clip_grad_norm_ (in place)
        for n, p in m.state_dict().items():
            d[f"step{s}.after.{n}"] = npy(p)
    np.savez_compressed(os.path.join(OUT, fname), **d)


def gen_state_dict_layout():
    """Key/shape layout of the reference's saved LR checkpoints (result/Criteo_x1/20241122/*/LR.model)."""
    p = os.path.join(REF, "result/Criteo_x1/20241122/140942/LR.model")
    sd = torch.load(p, map_location="cpu", weights_only=True)
    layout = {k: list(v.shape) for k, v in sd.items()}
    with open(os.path.join(OUT, "lr_state_dict_layout.json"), "w") as f:
        json.dump(layout, f, indent=1)


def gen_train_reg_sgd():
    """SGD with the L2 term on every nn.Embedding (BaseModel.py:214-231): every table row shrinks in every step.
    The third trace has a lambda large enough for lambda * ||W|| to exceed max_norm = 10, so clip_grad_norm_ bites and
    the shrink factor changes from step to step."""
    gen_train("lr", "SGD", 1e-2, "train_lr_sgd_reg.npz", scale_tables=1000.0, lr=0.05, steps=4)
    gen_train("deepfm", "SGD", 1e-3, "train_deepfm_sgd_reg.npz", hidden=(16, 8), scale_tables=1000.0, lr=0.05, steps=4)
    gen_train("fm", "SGD", 4.0, "train_fm_sgd_reg_clip.npz", scale_tables=1000.0, lr=0.02, steps=4)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    if sys.argv[1:] == ["reg_sgd"]:          # only the fixtures of gen_train_reg_sgd (the others stay as committed)
        gen_train_reg_sgd()
        sys.exit(0)
    gen_layers()
    gen_mlp()
    gen_interactions()
    gen_train("lr", "Adam", 5e-7, "train_lr_adam_reg.npz", scale_tables=1000.0)
    gen_train("fm", "SGD", None, "train_fm_sgd.npz", scale_tables=1000.0, lr=0.05)
    gen_train("deepfm", "Adam", None, "train_deepfm_adam.npz", hidden=(16, 8), scale_tables=1000.0)
    gen_train("deepfm", "SGD", None, "train_deepfm_sgd_bn.npz", hidden=(16, 8), normalization="bn",
              scale_tables=1000.0, lr=0.05)
    gen_train("deepfm", "AdamW", 1e-4, "train_deepfm_adamw_reg.npz", hidden=(16, 8), scale_tables=1000.0)
    # residual stacks (DeepCrossing): ReLU+BatchNorm under SGD, tanh+LayerNorm under Adam
    gen_train("deepcrossing", "SGD", None, "train_deepcrossing_sgd_bn.npz", hidden=(16, 8), normalization="bn",
              scale_tables=1000.0, lr=0.05)
    gen_train("deepcrossing", "Adam", None, "train_deepcrossing_adam_ln.npz", hidden=(16, 8), normalization="ln",
              scale_tables=1000.0, activation="tanh")
    # longer traces: rows stay absent from the batch for several steps in a row while their Adam moments still move them
    gen_train("deepfm", "Adam", None, "train_deepfm_adam_long.npz", hidden=(16, 8), scale_tables=1000.0, steps=12,
              bs=8, slim=True)
    gen_train("deepfm", "AdamW", None, "train_deepfm_adamw_long.npz", hidden=(16, 8), scale_tables=1000.0, steps=8,
              bs=8, slim=True)
    gen_train_reg_sgd()
    gen_state_dict_layout()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes:", tot, sorted(os.listdir(OUT)))
