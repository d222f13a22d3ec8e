"""Drop-in counterparts of the reference's model/utils/Layers.py on HIP kernels.

Same class names, constructor arguments, forward signatures, parameter names and state_dict
keys as the reference (so a BaseModel-style loop, its regulariser and its checkpoints work
unchanged); the arithmetic runs in librs_hip.so.

Storage: every Embedding_layer keeps ONE stacked table [sum_f V_f, d] (and one [sum_f V_f, 1]
for the 1-dim tables); the per-column ``nn.Embedding`` modules the reference exposes
(``embedded_features[col]``, ``embedded_to_1_dim_features[col]``) are views into it, so one
kernel launch serves all fields while ``state_dict()``/``parameters()`` look like the
reference's.
"""
from __future__ import annotations

import torch
from torch import nn

from . import functional as Fn
from . import ops
from ._lib import RSError
from .ops import ACT_CODES, FieldLayout, SparsePlan


class _StackedTables:
    """Keeps per-column nn.Embedding weights aliased to one stacked storage."""

    def __init__(self, modules: nn.ModuleDict, cols, dim):
        self.modules = modules
        self.cols = cols
        self.dim = dim
        self.vocab = [modules[c].num_embeddings for c in cols]
        self.stacked = None
        self.layout = None
        self._plans = {}
        self.version = 0            # bumped whenever `stacked` becomes a different tensor
        self.row_state = None       # ops.RowAdam of the trainer that steps this table on touched rows only

    def weights(self):
        return [self.modules[c].weight for c in self.cols]

    def restack(self):
        """Make every per-column weight a view of one stacked table.  On the device the table already lives on, the
        existing storage is kept (trainers, their optimizer state and captured HIP graphs hold its address): columns
        that were given storage of their own are copied back into it.  Only a move to another device allocates."""
        ws = self.weights()
        if not ws:
            return
        dev = ws[0].device
        st = self.stacked
        if st is not None and st.device == dev and all(w.device == dev for w in ws):
            base, esz, o = st.data_ptr(), st.element_size() * self.dim, 0
            with torch.no_grad():
                for w, v in zip(ws, self.vocab):
                    if w.data_ptr() != base + o * esz:
                        st[o:o + v].copy_(w.data)
                        w.data = st[o:o + v]
                    o += v
            return
        total = sum(self.vocab)
        stacked = torch.empty((total, self.dim), dtype=torch.float32, device=dev)
        o = 0
        with torch.no_grad():
            for w, v in zip(ws, self.vocab):
                stacked[o:o + v].copy_(w.data)
                w.data = stacked[o:o + v]
                o += v
        if st is not None:
            self.version += 1
            self.row_state = None   # the optimizer state stayed on the old device: its trainer must be rebuilt
        self.stacked = stacked
        if self.layout is None or self.layout.device != dev:
            self.layout = FieldLayout(self.vocab, dev) if dev.type == "cuda" else None
            self._plans = {}

    def before_read(self, idx, mark_ahead: int = 0):
        """Rows that a touched-rows Adam trainer left behind are brought up to the current step before `idx` reads
        them (no-op without such a trainer)."""
        if self.row_state is not None:
            self.row_state.catch_up(idx, self.layout, mark_ahead)

    def flush_rows(self):
        if self.row_state is not None:
            self.row_state.flush()

    def stacked_for(self, weights):
        """The stacked table, after checking that the given per-column weights still alias it."""
        st = self.stacked
        ok = st is not None and len(weights) == len(self.vocab)
        if ok:
            base, o, esz = st.data_ptr(), 0, st.element_size() * self.dim
            for w, v in zip(weights, self.vocab):
                if w.data_ptr() != base + o * esz or w.device != st.device:
                    ok = False
                    break
                o += v
        if not ok:
            self.restack()
            st = self.stacked
        if not st.is_cuda:
            raise RSError("Embedding_layer: parameters are on the CPU; move the model to the MI355X first "
                          "(this package has no CPU path)")
        return st

    def plan_for(self, B, d, with_lr=False) -> SparsePlan:
        key = (B, d, with_lr)
        if key not in self._plans:
            self._plans[key] = SparsePlan(self.layout, cap=B, d=d, with_lr=with_lr)
        return self._plans[key]


class Embedding_layer(nn.Module):
    """Reference: model/utils/Layers.py:112-167."""

    def __init__(self, dataRecorder):
        super().__init__()
        self.dataRecorder = dataRecorder
        self.embedding_dim = dataRecorder.embedding_dim
        self.feature_map = dataRecorder.feature_map
        embedding_dict = {}
        embedding_to_1_dim_dict = {}
        for col_name, col_info in self.feature_map.items():
            if col_info["type"] == "categorical":
                vocab_size = col_info["vocab_size"]
                # same construction order (hence the same RNG stream) as Layers.py:118-124
                embedding_dict[col_name] = self.initialize_embeddings(vocab_size, self.embedding_dim)
                embedding_to_1_dim_dict[col_name] = self.initialize_embeddings(vocab_size, 1)
        self.embedded_features = nn.ModuleDict(embedding_dict)
        self.embedded_to_1_dim_features = nn.ModuleDict(embedding_to_1_dim_dict)
        self._cols = self._lookup_columns()
        self._tab = _StackedTables(self.embedded_features, self._cols, self.embedding_dim)
        self._tab1 = _StackedTables(self.embedded_to_1_dim_features, self._cols, 1)
        self._tab.restack()
        self._tab1.restack()

    def _lookup_columns(self):
        cols = []
        for i in range(len(self.feature_map)):            # Layers.py:143-146
            column = self.dataRecorder.features[i]
            if self.feature_map[column]["type"] == "numerical":
                continue
            cols.append(column)
        return cols

    def initialize_embeddings(self, num_embeddings, emb_dim, pad_idx=0, init_std=1e-4):
        """Layers.py:130-135: default nn.Embedding, then rows 1.. ~ N(0, init_std^2); row 0 stays 0."""
        embedding = nn.Embedding(num_embeddings=num_embeddings, embedding_dim=emb_dim, padding_idx=pad_idx)
        torch.nn.init.normal_(embedding.weight[1:, :], mean=0.0, std=init_std)
        return embedding

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self._tab.restack()          # .to()/.cuda() gave every column its own storage: re-alias
        self._tab1.restack()
        return r

    @property
    def lookup_columns(self):
        return list(self._cols)

    def index_matrix(self, x) -> torch.Tensor:
        """dict of column tensors -> columnar [F, B] int64 matrix on the table's device."""
        cap = Fn.TableCapture.active
        if cap is not None:
            idx = cap.index_matrix(x, self._cols, lambda: Fn.stack_columns(x, self._cols))
        else:
            idx = Fn.stack_columns(x, self._cols)
        if not idx.is_cuda:
            raise RSError("Embedding_layer: batch tensors are on the CPU; move them to the device first "
                          "(BaseModel.features_to_divice does)")
        return idx

    def forward(self, x, scalar_only=False):
        """(B, n_categorical, d), or (B, n_categorical, 1) with scalar_only=True.

        Layers.py:137-160.  The reference's scalar_only branch stops in leftover debug code
        (Layers.py:155-157); this returns what its unreachable line 158 would."""
        idx = self.index_matrix(x)
        tab = self._tab1 if scalar_only else self._tab
        return Fn.embedding_lookup(idx, tab, tab.weights())

    def dict2tensor(self, tensor_dict):
        return torch.stack(list(tensor_dict.values()), dim=1)      # Layers.py:162-167


class LinearRegression_layer(nn.Module):
    """Reference: model/utils/Layers.py:97-109 (one-hot LR = sum_f w_f[idx_f] + b)."""

    def __init__(self, dataRecoder):
        super().__init__()
        self.dataRecoder = dataRecoder
        self.scalar_only = True
        self.scalar = Embedding_layer(self.dataRecoder)
        self.bias = nn.Parameter(torch.zeros(1), requires_grad=True)

    def forward(self, x):
        idx = self.scalar.index_matrix(x)
        tab1 = self.scalar._tab1
        return Fn.lr_forward(idx, tab1, self.bias, tab1.weights())


class Linear(nn.Linear):
    """nn.Linear (same parameters and state_dict keys) whose forward and backward run on the HIP GEMM;
    used where the reference's models hold a bare nn.Linear (model/DCN/DCN.py:31, model/NFM/NFM.py:18)."""

    def forward(self, x):
        x2d = x.reshape(x.shape[0], -1).contiguous()
        if not x2d.is_cuda:
            raise RSError("Linear: input is on the CPU (this package has no CPU path)")
        return Fn.mlp_block(x2d, self.weight, self.bias, 0, None, None, None, None, self.training, None, None)


class Dropout(nn.Dropout):
    """nn.Dropout (same constructor, no parameters) whose training-mode forward and backward run on the HIP kernel
    (Layers.py:33-37,46).  The masks come from a counter-based generator keyed by torch's seed, not from torch's
    generator: same distribution, different bits."""

    _salts = 0

    def __init__(self, p=0.5):
        super().__init__(p)
        Dropout._salts += 1
        # tells the masks of different layers apart; BaseModel.model_to_device() renumbers the layers of a model by
        # their position in it, the construction counter only serves layers used outside a model
        self.salt = Dropout._salts

    def forward(self, x):
        if not self.training or self.p == 0:
            return x
        return Fn.dropout(x, self.p, self.salt)


def _activation_module(name: str):
    name = name.lower()
    if name == "relu":
        return nn.ReLU(), ACT_CODES["relu"]
    if name == "prelu":
        return nn.PReLU(), ACT_CODES["prelu"]
    if name == "tanh":
        return nn.Tanh(), ACT_CODES["tanh"]
    if name == "leakyrelu":
        return nn.LeakyReLU(), ACT_CODES["leakyrelu"]
    raise ValueError(f"unknown activation {name!r}")


class MLPBlock(nn.Module):
    """Reference: model/utils/Layers.py:8-47.  Linear -> (BN | LN) -> activation -> dropout."""

    def __init__(self, input_dim, output_dim, activation_func="ReLU", normalization="bn", dropout_rate=None):
        super().__init__()
        self.fc = nn.Linear(input_dim, output_dim)
        self.activation_func, self._act = _activation_module(activation_func)
        self.normalization = normalization
        if not normalization:
            pass
        elif normalization == "bn":
            self.norm_layer = nn.BatchNorm1d(output_dim)
        elif normalization == "ln":
            self.norm_layer = nn.LayerNorm(output_dim)
        else:
            raise ValueError("normalization must be either 'bn' or 'ln' or None")
        self.dropout = Dropout(dropout_rate) if dropout_rate else Dropout(0)

    def _norm_args(self):
        if not self.normalization:
            return None, None, None, None, 1e-5
        n = self.norm_layer
        if self.normalization == "bn":
            if self.training and n.num_batches_tracked is not None:
                n.num_batches_tracked.add_(1)
            return n.weight, n.bias, n.running_mean, n.running_var, n.eps
        return n.weight, n.bias, None, None, n.eps

    def forward(self, x):
        bs = x.size()[0]
        x = x.reshape(bs, -1)
        gamma, beta, rm, rv, eps = self._norm_args()
        slope = self.activation_func.weight if self._act == ACT_CODES["prelu"] else None
        out = Fn.mlp_block(x, self.fc.weight, self.fc.bias, self._act, self.normalization or None, gamma, beta, slope,
                           self.training, rm, rv, eps)
        return self.dropout(out)


class MLPStack(nn.Sequential):
    """nn.Sequential of MLPBlocks (what the reference's models build: model/DeepFM/DeepFM.py:25-38) whose forward hands
    the whole run to one autograd node, so that its backward can fuse across block boundaries (functional._MLPStackFn).
    Same children, same state_dict keys as the nn.Sequential; anything that is not a run of MLPBlocks on the GPU goes
    block by block."""

    def forward(self, x):
        blocks = list(self)
        if len(blocks) < 2 or not all(type(b) is MLPBlock for b in blocks) or not x.is_cuda:
            return super().forward(x)
        x = x.reshape(x.size()[0], -1)
        if not x.is_contiguous():
            x = x.contiguous()
        return Fn.mlp_stack(x, blocks)


class ResidualMLPBlock(nn.Module):
    """Reference: model/utils/Layers.py:50-94.  identity (or downsample(x)) + block(x)."""

    def __init__(self, input_dim, output_dim, activation_func, normalization="bn", dropout_rate=None):
        super().__init__()
        self.fc = nn.Linear(input_dim, output_dim)
        if activation_func.lower() == "prelu":
            # the reference stores the nn.PReLU class itself (Layers.py:60) and fails at forward
            raise ValueError("ResidualMLPBlock: 'prelu' is not usable in the reference either; "
                             "use relu / tanh / leakyrelu")
        self.activation_func, self._act = _activation_module(activation_func)
        self.normalization = normalization
        if normalization == "bn":
            self.norm_layer = nn.BatchNorm1d(output_dim)
        elif normalization == "ln":
            self.norm_layer = nn.LayerNorm(output_dim)
        else:
            raise ValueError("normalization must be either 'bn' or 'ln'")
        self.dropout = Dropout(dropout_rate) if dropout_rate else Dropout(0)
        self.downsample = nn.Linear(input_dim, output_dim) if input_dim != output_dim else None

    _norm_args = MLPBlock._norm_args

    def forward(self, x):
        bs = x.size()[0]
        x = x.reshape(bs, -1)
        gamma, beta, rm, rv, eps = self._norm_args()
        Wd = self.downsample.weight if self.downsample is not None else None
        bd = self.downsample.bias if self.downsample is not None else None
        if self.dropout.p and self.training:
            # dropout sits between the block and the residual add (Layers.py:88-94): the three pieces one by one
            out = Fn.mlp_block(x, self.fc.weight, self.fc.bias, self._act, self.normalization, gamma, beta, None,
                               self.training, rm, rv, eps)
            out = self.dropout(out)
            identity = x if Wd is None else Fn.mlp_block(x, Wd, bd, 0, None, None, None, None, self.training, None, None)
            return Fn.add(identity, out)
        return Fn.residual_block(x, self.fc.weight, self.fc.bias, gamma, beta, Wd, bd, self._act, self.normalization,
                                 self.training, rm, rv, eps)
