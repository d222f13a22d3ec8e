"""CPU oracle for the CTR training hot path.  TEST INFRASTRUCTURE ONLY.

This module restates, in stock torch CPU ops, the arithmetic of the reference's
hot path so that the HIP kernels can be checked against it.  It is never
imported by the product package; only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may use it.

Parity status: PINNED.  Every function below is checked against golden vectors
captured from the reference's own Python (oracle/gen_golden.py imports
/root/reference and writes tests/golden/*.npz; tests/test_oracle_golden.py
replays them).

Each function cites the reference lines it follows (paths relative to the
reference root).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# feature bookkeeping
# --------------------------------------------------------------------------
def categorical_columns(feature_map: Dict[str, dict], features: Sequence[str]) -> List[str]:
    """Columns the embedding layer actually embeds, in lookup order.

    model/utils/Layers.py:143-146 walks ``features`` for ``len(feature_map)``
    entries and skips numerical columns.
    """
    cols = []
    for i in range(len(feature_map)):
        c = features[i]
        if feature_map[c]["type"] == "numerical":
            continue
        cols.append(c)
    return cols


def init_table(vocab_size: int, dim: int, init_std: float = 1e-4,
               generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """model/utils/Layers.py:130-135: rows 1.. ~ N(0, std^2), row 0 (padding) = 0."""
    w = torch.zeros(vocab_size, dim)
    w[1:].normal_(0.0, init_std, generator=generator)
    return w


# --------------------------------------------------------------------------
# embedding lookup / LR
# --------------------------------------------------------------------------
def embedding_lookup(tables: Dict[str, torch.Tensor], cols: Sequence[str],
                     x: Dict[str, torch.Tensor]) -> torch.Tensor:
    """model/utils/Layers.py:147-153,162-167: per-column row gather, stacked on dim 1."""
    return torch.stack([tables[c][x[c].long()] for c in cols], dim=1)


def lr_forward(tables1: Dict[str, torch.Tensor], bias: torch.Tensor, cols: Sequence[str],
               x: Dict[str, torch.Tensor]) -> torch.Tensor:
    """model/utils/Layers.py:107-109 with the intended body of :150-151,:158.

    ``stack(1-dim lookups, dim=1).sum(dim=1) + bias`` -> (B, 1).
    """
    rows = torch.stack([tables1[c][x[c].long()] for c in cols], dim=1)  # (B, F, 1)
    return rows.sum(dim=1) + bias


# --------------------------------------------------------------------------
# interactions
# --------------------------------------------------------------------------
def fm_second_order(emb: torch.Tensor) -> torch.Tensor:
    """model/FM/FM.py:27-30 (same lines in model/DeepFM/DeepFM.py:47-49)."""
    square_of_sum = emb.sum(dim=1).pow(2)
    sum_of_square = emb.pow(2).sum(dim=1)
    return 0.5 * (square_of_sum - sum_of_square).sum(dim=1, keepdim=True)


def bi_interaction_textbook(emb: torch.Tensor) -> torch.Tensor:
    """Textbook NFM Bi-Interaction pooling: 0.5*((sum_f e)^2 - sum_f e^2) -> (B, d)."""
    return 0.5 * (emb.sum(dim=1).pow(2) - emb.pow(2).sum(dim=1))


def nfm_bi_interaction_reference(emb_flat: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """model/NFM/NFM.py:42-44 on flattened (B, F*d) embeddings, v is (F*d, k)."""
    square_of_sum = (emb_flat @ v).pow(2)
    sum_of_square = emb_flat.pow(2) @ v.pow(2)
    return 0.5 * (square_of_sum - sum_of_square)


def _pairs(n: int):
    return [(i, j) for i in range(n) for j in range(i + 1, n)]


def ffm_interaction(ff_tables: Sequence[Sequence[torch.Tensor]], x: torch.Tensor,
                    padding_idx: Optional[int] = None) -> torch.Tensor:
    """model/FFM/FFM.py:43-62 with independent tables.

    ``ff_tables[i][j]`` is the table of feature i as seen by field j, (V_i, d);
    x is (B, F) long.  Accumulates pair by pair in (i, j) row-major order.
    padding_idx: the lookups are nn.Embedding modules in the reference; with the padding_idx=0 of
    model/utils/Layers.py:132 that row receives no gradient.
    """
    bs, nf = x.shape
    out = torch.zeros(bs, 1)
    for i, j in _pairs(nf):
        vi_fj = F.embedding(x[:, i], ff_tables[i][j], padding_idx=padding_idx)
        vj_fi = F.embedding(x[:, j], ff_tables[j][i], padding_idx=padding_idx)
        out = out + (vi_fj * vj_fi).sum(dim=-1, keepdim=True)
    return out


def pnn_inner(emb_flat: torch.Tensor, nf: int, d: int) -> torch.Tensor:
    """model/PNN/PNN.py:65-77: <e_i, e_j> for i<j, row-major pair order -> (B, P)."""
    e = emb_flat.view(emb_flat.shape[0], nf, d)
    return torch.cat([(e[:, i] * e[:, j]).sum(dim=1, keepdim=True) for i, j in _pairs(nf)], dim=1)


def pnn_outer(emb_flat: torch.Tensor, nf: int, d: int) -> torch.Tensor:
    """model/PNN/PNN.py:80-103: e_i e_j^T flattened (i index major) -> (B, P*d*d)."""
    bs = emb_flat.shape[0]
    e = emb_flat.view(bs, nf, d)
    return torch.cat([(e[:, i].unsqueeze(2) * e[:, j].unsqueeze(1)).reshape(bs, -1)
                      for i, j in _pairs(nf)], dim=1)


def afm_pairwise(emb: torch.Tensor) -> torch.Tensor:
    """model/AFM/AFM.py:47-56: Hadamard products of all field pairs -> (B, P, d)."""
    nf = emb.shape[1]
    return torch.stack([emb[:, i] * emb[:, j] for i, j in _pairs(nf)], dim=1)


def afm_forward(emb: torch.Tensor, att_w: torch.Tensor, att_b: torch.Tensor, att_h: torch.Tensor,
                p_w: torch.Tensor, p_b: torch.Tensor) -> torch.Tensor:
    """model/AFM/AFM.py:19-24,36-42.

    att_w (A, d), att_b (A,), att_h (1, A) [no bias], p_w (1, d), p_b (1,).
    Returns the AFM term (B, 1) (without the LR term and without sigmoid).
    """
    pw = afm_pairwise(emb)                                   # (B, P, d)
    s = F.relu(pw @ att_w.t() + att_b) @ att_h.t()           # (B, P, 1)
    a = torch.softmax(s, dim=1)
    pooled = (a * pw).sum(dim=1)                             # (B, d)
    return pooled @ p_w.t() + p_b


def dcn_cross(x0: torch.Tensor, ws: Sequence[torch.Tensor], bs: Sequence[torch.Tensor]) -> torch.Tensor:
    """model/DCN/DCN.py:48-52 in its intended shape: x0 (B, D), w_l (D, 1), b_l (D,) or (1, D)."""
    x = x0
    for w, b in zip(ws, bs):
        x = x0 * (x @ w.reshape(-1, 1)) + b.reshape(1, -1) + x
    return x


# --------------------------------------------------------------------------
# MLP blocks
# --------------------------------------------------------------------------
def _activation(name: str, z: torch.Tensor, prelu_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    name = name.lower()
    if name == "relu":
        return F.relu(z)
    if name == "tanh":
        return torch.tanh(z)
    if name == "leakyrelu":
        return F.leaky_relu(z, 0.01)
    if name == "prelu":
        return F.prelu(z, prelu_weight)
    raise ValueError(name)


def _normalize(kind, z, gamma, beta, training=True, running_mean=None, running_var=None, eps=1e-5):
    if not kind:
        return z
    if kind == "bn":
        return F.batch_norm(z, running_mean, running_var, gamma, beta, training, 0.1, eps)
    if kind == "ln":
        return F.layer_norm(z, (z.shape[-1],), gamma, beta, eps)
    raise ValueError("normalization must be either 'bn' or 'ln' or None")


def mlp_block(x, weight, bias, activation="ReLU", normalization=None, gamma=None, beta=None,
              prelu_weight=None, training=True, running_mean=None, running_var=None):
    """model/utils/Layers.py:39-47 (dropout 0): act(norm(x.view(B,-1) @ W^T + b))."""
    x = x.reshape(x.shape[0], -1)
    z = F.linear(x, weight, bias)
    z = _normalize(normalization, z, gamma, beta, training, running_mean, running_var)
    return _activation(activation, z, prelu_weight)


def residual_mlp_block(x, weight, bias, activation, normalization, gamma, beta,
                       down_weight=None, down_bias=None, training=True,
                       running_mean=None, running_var=None):
    """model/utils/Layers.py:84-94: identity (or downsample(x)) + act(norm(fc(x)))."""
    x = x.reshape(x.shape[0], -1)
    out = F.linear(x, weight, bias)
    out = _normalize(normalization, out, gamma, beta, training, running_mean, running_var)
    out = _activation(activation, out)
    identity = x if down_weight is None else F.linear(x, down_weight, down_bias)
    return identity + out


# --------------------------------------------------------------------------
# whole-model oracle with autograd: the training step
# --------------------------------------------------------------------------
class OracleMLPBlock(nn.Module):
    """Same parameters and forward as model/utils/Layers.py:8-47 (dropout 0)."""

    def __init__(self, in_dim, out_dim, activation="ReLU", normalization=None):
        super().__init__()
        self.fc = nn.Linear(in_dim, out_dim)
        self.activation = activation.lower()
        self.normalization = normalization
        if self.activation == "prelu":
            self.prelu = nn.PReLU()
        if normalization == "bn":
            self.norm_layer = nn.BatchNorm1d(out_dim)
        elif normalization == "ln":
            self.norm_layer = nn.LayerNorm(out_dim)
        elif normalization:
            raise ValueError("normalization must be either 'bn' or 'ln' or None")

    def forward(self, x):
        z = self.fc(x.reshape(x.shape[0], -1))
        if self.normalization:
            z = self.norm_layer(z)
        pw = self.prelu.weight if self.activation == "prelu" else None
        return _activation(self.activation, z, pw)


class OracleResidualBlock(nn.Module):
    """Same parameters and forward as model/utils/Layers.py:50-94 (dropout 0): identity (or downsample(x)) +
    act(norm(fc(x))); normalisation is mandatory there (Layers.py:68-73)."""

    def __init__(self, in_dim, out_dim, activation="ReLU", normalization="bn"):
        super().__init__()
        self.fc = nn.Linear(in_dim, out_dim)
        self.activation = activation.lower()
        self.normalization = normalization
        if normalization == "bn":
            self.norm_layer = nn.BatchNorm1d(out_dim)
        elif normalization == "ln":
            self.norm_layer = nn.LayerNorm(out_dim)
        else:
            raise ValueError("normalization must be either 'bn' or 'ln'")
        self.downsample = nn.Linear(in_dim, out_dim) if in_dim != out_dim else None

    def forward(self, x):
        x = x.reshape(x.shape[0], -1)
        out = _activation(self.activation, self.norm_layer(self.fc(x)))
        identity = x if self.downsample is None else self.downsample(x)
        return identity + out


class OracleCTRModel(nn.Module):
    """Per-field nn.Embedding tables + one of the reference's model heads.

    kind: "lr" | "fm" | "deepfm" | "dnn" | "widedeep" | "deepcrossing" | "ffm" | "pnn" | "nfm" | "afm" | "dcn".
    Outputs LOGITS (B, 1);
    the training step applies sigmoid before BCE the way model/LR/LR.py:29-33
    does.  Embedding tables follow model/utils/Layers.py:118-135 (one
    nn.Embedding(V, d, padding_idx=0) and one nn.Embedding(V, 1, padding_idx=0)
    per categorical column).
    """

    def __init__(self, kind, feature_map, features, dim, hidden=(), activation="ReLU",
                 normalization=None, init_std=1e-4, mirror_unused=False, pnn_type="inner", nfm_hidden_dim=4,
                 attention_dim=8, cross_layers_num=3):
        super().__init__()
        self.kind = kind
        self.pnn_type = pnn_type
        self.cols = categorical_columns(feature_map, features)
        self.dim = dim
        self.emb = nn.ModuleDict()
        self.emb1 = nn.ModuleDict()
        # model/DeepCrossing/DeepCrossing.py has no LinearRegression_layer: its only tables are BaseModel's
        # embedding_layer (BaseModel.py:47)
        self.has_lr_tables = kind != "deepcrossing"
        for c in self.cols:
            v = feature_map[c]["vocab_size"]
            e = nn.Embedding(v, dim, padding_idx=0)
            e1 = nn.Embedding(v, 1, padding_idx=0)
            with torch.no_grad():
                e.weight[1:].normal_(0, init_std)
                e1.weight[1:].normal_(0, init_std)
                e.weight[0].zero_()
                e1.weight[0].zero_()
            self.emb[c] = e
            if self.has_lr_tables:
                self.emb1[c] = e1
        if mirror_unused:
            # The reference builds two Embedding_layer objects per model (BaseModel.py:47 and
            # Layers.py:102), each with d-dim AND 1-dim tables; half of them never see a batch
            # but they are parameters: L2 regularisation and clip_grad_norm_ include them.
            if self.has_lr_tables:
                self.unused_emb = nn.ModuleDict({c: nn.Embedding(feature_map[c]["vocab_size"], dim, padding_idx=0)
                                                 for c in self.cols})
            self.unused_emb1 = nn.ModuleDict({c: nn.Embedding(feature_map[c]["vocab_size"], 1, padding_idx=0)
                                              for c in self.cols})
        if self.has_lr_tables:
            self.bias = nn.Parameter(torch.zeros(1))
        if kind == "deepcrossing":                  # model/DeepCrossing/DeepCrossing.py:15-26
            dims = [len(self.cols) * dim] + list(hidden) + [1]
            self.mlp = nn.Sequential(*[OracleResidualBlock(dims[i], dims[i + 1], activation, normalization)
                                       for i in range(len(dims) - 1)])
            for m in self.mlp.modules():            # model/utils/BaseModel.py:79-91
                if type(m) is nn.Linear:
                    nn.init.xavier_normal_(m.weight)
                    m.bias.data.fill_(0)
        if kind in ("deepfm", "dnn", "widedeep"):
            dims = [len(self.cols) * dim] + list(hidden) + [1]
            # model/DeepFM/DeepFM.py:25-38: every layer, the 1-wide head included, is an MLPBlock
            self.deep = nn.Sequential(*[OracleMLPBlock(dims[i], dims[i + 1], activation, normalization)
                                        for i in range(len(dims) - 1)])
            for m in self.deep.modules():           # model/utils/BaseModel.py:79-91
                if type(m) is nn.Linear:
                    nn.init.xavier_normal_(m.weight)
                    m.bias.data.fill_(0)

        nf, D = len(self.cols), len(self.cols) * dim

        def stack(dims):
            blocks = nn.Sequential(*[OracleMLPBlock(dims[i], dims[i + 1], activation, normalization)
                                     for i in range(len(dims) - 1)])
            for m in blocks.modules():
                if type(m) is nn.Linear:
                    nn.init.xavier_normal_(m.weight)
                    m.bias.data.fill_(0)
            return blocks

        if kind == "ffm":                           # model/FFM/FFM.py:21-23 with independent tables
            self.ffm = nn.ModuleDict()
            for c in self.cols:
                e = nn.Embedding(feature_map[c]["vocab_size"], nf * dim, padding_idx=0)
                with torch.no_grad():
                    e.weight[1:].normal_(0, init_std)
                self.ffm[c] = e
        elif kind == "pnn":                         # model/PNN/PNN.py:19-40
            pairs = nf * (nf - 1) // 2
            pnn_dim = {"inner": pairs, "outer": pairs * dim * dim, "both": pairs + pairs * dim * dim}[pnn_type]
            self.deep = stack([D + pnn_dim] + list(hidden) + [1])
        elif kind == "nfm":                         # model/NFM/NFM.py:16-33
            self.v = nn.Parameter(torch.randn(D, nfm_hidden_dim))
            self.linear = nn.Linear(nf, 1)
            nn.init.xavier_normal_(self.linear.weight)
            self.linear.bias.data.fill_(0)
            self.deep = stack([nfm_hidden_dim] + list(hidden) + [1])
        elif kind == "afm":                         # model/AFM/AFM.py:19-26
            self.attention = nn.Sequential(nn.Linear(dim, attention_dim), nn.ReLU(),
                                           nn.Linear(attention_dim, 1, bias=False), nn.Softmax(dim=1))
            self.p = nn.Linear(dim, 1)
        elif kind == "dcn":                         # model/DCN/DCN.py:16-31
            self.cross_w = nn.ParameterList([nn.Parameter(torch.randn(D, 1)) for _ in range(cross_layers_num)])
            self.cross_b = nn.ParameterList([nn.Parameter(torch.randn(D, 1)) for _ in range(cross_layers_num)])
            self.deep = stack([D] + list(hidden))
            self.fc = nn.Linear(D + list(hidden)[-1], 1)

    def embed(self, x):
        return torch.stack([self.emb[c](x[c].long()) for c in self.cols], dim=1)

    def lr(self, x):
        return torch.stack([self.emb1[c](x[c].long()) for c in self.cols], dim=1).sum(dim=1) + self.bias

    def forward(self, x):
        if self.kind == "lr":
            return self.lr(x)
        e = self.embed(x)
        if self.kind == "fm":                       # model/FM/FM.py:19-33
            return self.lr(x) + fm_second_order(e)
        if self.kind == "deepfm":                   # model/DeepFM/DeepFM.py:42-55
            return self.lr(x) + fm_second_order(e) + self.deep(e)
        if self.kind == "dnn":                      # model/DNN/DNN.py:40-44
            return self.deep(e)
        if self.kind == "widedeep":                 # model/WideDeep/WideDeep.py:43-51
            return self.lr(x) + self.deep(e)
        if self.kind == "deepcrossing":             # model/DeepCrossing/DeepCrossing.py:32-36
            return self.mlp(e)
        B, nf, d = e.shape
        if self.kind == "ffm":                      # model/FFM/FFM.py:32-40
            xm = torch.stack([x[c].long() for c in self.cols], dim=1)
            tabs = [[self.ffm[c].weight[:, j * d:(j + 1) * d] for j in range(nf)] for c in self.cols]
            return self.lr(x) + ffm_interaction(tabs, xm, padding_idx=0)
        if self.kind == "pnn":                      # model/PNN/PNN.py:45-60
            flat = e.view(B, -1)
            parts = []
            if self.pnn_type in ("inner", "both"):
                parts.append(pnn_inner(flat, nf, d))
            if self.pnn_type in ("outer", "both"):
                parts.append(pnn_outer(flat, nf, d))
            return self.deep(torch.cat([flat] + parts, dim=1))
        if self.kind == "nfm":                      # model/NFM/NFM.py:37-49
            xm = torch.stack([x[c].long() for c in self.cols], dim=1)
            return self.linear(xm.float()) + self.deep(nfm_bi_interaction_reference(e.view(B, -1), self.v))
        if self.kind == "afm":                      # model/AFM/AFM.py:32-45
            pw = afm_pairwise(e)
            att = self.attention(pw)
            return self.lr(x) + self.p(torch.sum(att * pw, dim=1))
        if self.kind == "dcn":                      # model/DCN/DCN.py:35-52 (bias broadcast over the batch)
            flat = e.view(B, -1)
            cross = dcn_cross(flat, list(self.cross_w), [b.view(-1) for b in self.cross_b])
            return self.fc(torch.cat([cross, self.deep(e)], dim=1))
        raise ValueError(self.kind)


def regularization_loss(model: nn.Module, emb_regular, net_regular):
    """model/utils/BaseModel.py:214-231: (lambda/2)*||p||_2^2 over weight/bias of every module;
    nn.Embedding params use emb_regular, everything else net_regular."""
    reg = 0
    if not (net_regular or emb_regular):
        return reg
    for _, module in model.named_modules():
        for p_name, p in module.named_parameters():
            if not p.requires_grad or p_name not in ("weight", "bias"):
                continue
            if type(module) is nn.Embedding:
                if emb_regular:
                    reg = reg + (emb_regular / 2) * torch.norm(p, 2) ** 2
            elif net_regular:
                reg = reg + (net_regular / 2) * torch.norm(p, 2) ** 2
    return reg


def make_optimizer(kind: str, params, lr: float):
    """model/utils/BaseModel.py:64-72 (torch defaults everywhere else)."""
    if kind == "SGD":
        return torch.optim.SGD(params, lr=lr)
    if kind == "Adam":
        return torch.optim.Adam(params, lr=lr)
    if kind == "AdamW":
        return torch.optim.AdamW(params, lr=lr)
    if kind == "Adagrad":            # not in the reference; the sparse fast path offers it
        return torch.optim.Adagrad(params, lr=lr)
    raise ValueError("Invalid value for optimizer")


def train_step(model: nn.Module, optimizer, x, labels, emb_regular=None, net_regular=None,
               max_norm: float = 10.0):
    """model/utils/BaseModel.py:141-150 with sigmoid on the logits (model/LR/LR.py:32).

    Returns (loss, total_norm_before_clip)."""
    optimizer.zero_grad()
    prob = torch.sigmoid(model(x))
    loss = F.binary_cross_entropy(prob.squeeze(1), labels.float())
    loss = loss + regularization_loss(model, emb_regular, net_regular)
    loss.backward()
    total_norm = nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
    optimizer.step()
    return loss.detach(), total_norm.detach()


def auc(labels: torch.Tensor, scores: torch.Tensor) -> float:
    """Rank-based ROC AUC with average ranks for ties (what sklearn.roc_auc_score,
    used at model/utils/BaseModel.py:171, computes)."""
    y = labels.double().flatten()
    s = scores.double().flatten()
    order = torch.argsort(s)
    s_sorted = s[order]
    ranks = torch.empty_like(s)
    n = s.numel()
    # average ranks over tie groups
    uniq, inv, counts = torch.unique_consecutive(s_sorted, return_inverse=True, return_counts=True)
    ends = torch.cumsum(counts, 0).double()
    starts = ends - counts.double() + 1
    avg = (starts + ends) / 2
    ranks[order] = avg[inv]
    n_pos = y.sum()
    n_neg = n - n_pos
    return float((ranks[y == 1].sum() - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg))


def logloss(labels: torch.Tensor, prob: torch.Tensor, eps: float = 1e-7) -> float:
    """model/utils/BaseModel.py:170 (sklearn log_loss with eps clipping)."""
    p = prob.double().clamp(eps, 1 - eps)
    y = labels.double()
    return float(-(y * p.log() + (1 - y) * (1 - p).log()).mean())
