"""Model classes over the HIP layers.

Constructors follow the reference's one live model script (model/LR/LR.py:10-26:
``Model(dataRecorder, lr, gpu, regular=None, ...)``); the other model files in the reference are
written against a removed DataRecorder API, so their forward bodies are taken as the specification
of the arithmetic and the constructor convention of LR.py is used for all of them.

Every model has ``logits(x)`` (parity is checked on these) and ``forward(x) = sigmoid(logits(x))``,
which is what BaseModel.train_step's binary_cross_entropy needs (LR.py:32).
"""
from __future__ import annotations

import torch
from torch import nn

from . import functional as Fn
from .base_model import BaseModel
from . import interactions as X
from .layers import Linear, LinearRegression_layer, MLPBlock, MLPStack, ResidualMLPBlock


def _mlp_stack(input_dim, hidden_dim_list, activation_func, normalization, dropout_rate, head=True, block=MLPBlock):
    """First block, hidden blocks, then (optionally) the 1-wide block -- itself a full block with
    normalisation and activation, as in model/DeepFM/DeepFM.py:25-38."""
    if block is MLPBlock:
        layers = [MLPBlock(input_dim, hidden_dim_list[0], activation_func, normalization, dropout_rate=dropout_rate)]
    else:
        layers = [block(input_dim, hidden_dim_list[0], activation_func, normalization)]
    for i in range(len(hidden_dim_list) - 1):
        layers.append(block(hidden_dim_list[i], hidden_dim_list[i + 1], activation_func, normalization))
    if head:
        layers.append(block(hidden_dim_list[-1], 1, activation_func, normalization))
    return MLPStack(*layers)


class _CTRModel(BaseModel):
    KIND = None

    def __init__(self, name, dataRecorder, lr, gpu, regular=None, optimizer_type="Adam", num_epochs=10, **kw):
        super().__init__(model_name=name, num_epochs=num_epochs, learning_rate=lr, record=kw.pop("record", False),
                         dataRecorder=dataRecorder, gpu=gpu, emb_regular=regular, net_regular=regular,
                         optimizer_type=optimizer_type, **kw)

    def _finish(self):
        self.initialize_criterion()
        self.initialize_optimizer()
        self.reset_parameters()
        self.model_to_device()

    def embedded_field_count(self):
        return len(self.embedding_layer.lookup_columns)

    def forward(self, x):
        return torch.sigmoid(self.logits(x))


class LR(_CTRModel):
    """model/LR/LR.py:10-33."""
    KIND = "lr"

    def __init__(self, dataRecorder, lr, gpu, regular=None, **kw):
        super().__init__("LR", dataRecorder, lr, gpu, regular, **kw)
        self.lr_layer = LinearRegression_layer(self.dataRecorder)
        self._finish()

    def logits(self, x):
        x = self.features_to_divice(x)
        return self.lr_layer(x)


class FM(_CTRModel):
    """model/FM/FM.py:19-33: LR + 0.5*sum_k((sum_f e)^2 - sum_f e^2)."""
    KIND = "fm"

    def __init__(self, dataRecorder, lr, gpu, regular=None, **kw):
        super().__init__("FM", dataRecorder, lr, gpu, regular, **kw)
        self.lr_layer = LinearRegression_layer(self.dataRecorder)
        self._finish()

    def logits(self, x):
        x = self.features_to_divice(x)
        first_order_output = self.lr_layer(x)
        emb_out = self.embedding_layer(x)
        return first_order_output + Fn.fm_second_order(emb_out)


class DeepFM(_CTRModel):
    """model/DeepFM/DeepFM.py:25-55."""
    KIND = "deepfm"

    def __init__(self, dataRecorder, lr, gpu, hidden_dim_list=(1024, 512, 256), activation_func="ReLU",
                 normalization=None, dropout_rate=0, regular=None, **kw):
        super().__init__("DeepFM", dataRecorder, lr, gpu, regular, **kw)
        self.lr_layer = LinearRegression_layer(self.dataRecorder)
        self.hidden_dim_list = list(hidden_dim_list)
        self.normalization = normalization
        self.dropout_rate = dropout_rate
        self.activation_func = activation_func
        in_dim = self.embedded_field_count() * self.dataRecorder.embedding_dim
        self.deep = _mlp_stack(in_dim, self.hidden_dim_list, activation_func, normalization, dropout_rate)
        self._finish()

    def logits(self, x):
        x = self.features_to_divice(x)
        first_order_output = self.lr_layer(x)
        emb_out = self.embedding_layer(x)
        fm_output = first_order_output + Fn.fm_second_order(emb_out)
        return fm_output + self.deep(emb_out)


class DNN(_CTRModel):
    """model/DNN/DNN.py:23-44."""
    KIND = "dnn"

    def __init__(self, dataRecorder, lr, gpu, hidden_dim_list=(512, 256), activation_func="ReLU", normalization="bn",
                 dropout_rate=0, regular=None, **kw):
        super().__init__("DNN", dataRecorder, lr, gpu, regular, **kw)
        in_dim = self.embedded_field_count() * self.dataRecorder.embedding_dim
        self.hidden_dim_list = list(hidden_dim_list)
        self.normalization = normalization
        self.activation_func = activation_func
        self.mlp = _mlp_stack(in_dim, self.hidden_dim_list, activation_func, normalization, dropout_rate)
        self._finish()

    def logits(self, x):
        x = self.features_to_divice(x)
        return self.mlp(self.embedding_layer(x))


class WideDeep(_CTRModel):
    """model/WideDeep/WideDeep.py:24-51."""
    KIND = "widedeep"

    def __init__(self, dataRecorder, lr, gpu, hidden_dim_list=(512, 256), activation_func="ReLU", normalization="bn",
                 dropout_rate=0, regular=None, **kw):
        super().__init__("WideDeep", dataRecorder, lr, gpu, regular, **kw)
        self.wide = LinearRegression_layer(self.dataRecorder)
        in_dim = self.embedded_field_count() * self.dataRecorder.embedding_dim
        self.hidden_dim_list = list(hidden_dim_list)
        self.normalization = normalization
        self.activation_func = activation_func
        self.deep = _mlp_stack(in_dim, self.hidden_dim_list, activation_func, normalization, dropout_rate)
        self._finish()

    def logits(self, x):
        x = self.features_to_divice(x)
        return self.wide(x) + self.deep(self.embedding_layer(x))


class DeepCrossing(_CTRModel):
    """model/DeepCrossing/DeepCrossing.py:15-36: a stack of residual blocks."""
    KIND = "deepcrossing"

    def __init__(self, dataRecorder, lr, gpu, hidden_dim_list=(1024, 512, 256), activation_func="ReLU",
                 normalization="bn", regular=None, **kw):
        super().__init__("DeepCrossing", dataRecorder, lr, gpu, regular, **kw)
        in_dim = self.embedded_field_count() * self.dataRecorder.embedding_dim
        self.mlp = _mlp_stack(in_dim, list(hidden_dim_list), activation_func, normalization, 0,
                              block=ResidualMLPBlock)
        self._finish()

    def logits(self, x):
        x = self.features_to_divice(x)
        return self.mlp(self.embedding_layer(x))


class FFM(_CTRModel):
    """model/FFM/FFM.py:32-62 with one independent table per (feature, partner field): LR + sum_{i<j}
    <v_{i,f_j}, v_{j,f_i}>."""
    KIND = "ffm"

    def __init__(self, dataRecorder, lr, gpu, regular=None, optimizer_type="AdamW", **kw):
        super().__init__("FFM", dataRecorder, lr, gpu, regular, optimizer_type=optimizer_type, **kw)
        self.linear = LinearRegression_layer(self.dataRecorder)
        self.field_aware = X.FieldAwareTables(self.dataRecorder.feature_map, self.embedding_layer.lookup_columns,
                                              self.dataRecorder.embedding_dim)
        self._finish()

    def logits(self, x):
        x = self.features_to_divice(x)
        idx = self.embedding_layer.index_matrix(x)
        return self.linear(x) + self.field_aware(idx)


class PNN(_CTRModel):
    """model/PNN/PNN.py:19-60: MLP over [embeddings | inner and/or outer products of all field pairs]."""
    KIND = "pnn"

    def __init__(self, dataRecorder, lr, gpu, hidden_dim_list=(1024, 512, 256), pnn_type="inner", activation_func="ReLU",
                 normalization="bn", dropout_rate=0, regular=None, **kw):
        super().__init__("PNN", dataRecorder, lr, gpu, regular, **kw)
        nf, d = self.embedded_field_count(), self.dataRecorder.embedding_dim
        pairs = nf * (nf - 1) // 2
        if pnn_type == "inner":
            pnn_dim = pairs
        elif pnn_type == "outer":
            pnn_dim = pairs * d * d
        elif pnn_type == "both":
            pnn_dim = pairs + pairs * d * d
        else:
            raise ValueError("pnn_type must be either 'inner' or 'outer' or 'both")
        self.pnn_type = pnn_type
        self.mlp = _mlp_stack(nf * d + pnn_dim, list(hidden_dim_list), activation_func, normalization, dropout_rate)
        self._finish()

    def inner_product(self, emb_out):
        return X.pnn_inner(emb_out)

    def outer_product(self, emb_out):
        return X.pnn_outer(emb_out)

    def both_product(self, emb_out):
        return torch.cat([X.pnn_inner(emb_out), X.pnn_outer(emb_out)], dim=1)

    def logits(self, x):
        x = self.features_to_divice(x)
        emb_out = self.embedding_layer(x)
        flat = emb_out.view(emb_out.shape[0], -1)
        if self.pnn_type == "inner":
            # one launch writes [embeddings | inner products | zeros up to a multiple of four columns]; the first block
            # takes the padded rows as they are (functional.mlp_block_fwd)
            xin = Fn.pnn_inner_input(emb_out)
            if xin is None:
                xin = torch.cat([flat, self.inner_product(emb_out)], dim=1)
            return self.mlp(xin)
        # The outer products are 741 * 256 values per sample at F = 39, d = 16: the first block takes them a block of
        # pairs at a time (functional.OuterLinear) instead of reading cat([flat, pnn_output]) (PNN.py:58); the columns
        # of its weight keep the reference's order [embeddings | inner | outer].
        other = flat if self.pnn_type == "outer" else torch.cat([flat, self.inner_product(emb_out)], dim=1)
        return self.mlp[1:](Fn.pnn_outer_block(self.mlp[0], emb_out, other))


class NFM(_CTRModel):
    """model/NFM/NFM.py:16-49 as written: a Linear over the raw (float) index values, the
    0.5*((E v)^2 - E^2 v^2) pooling with v ~ randn(F*d, k), then the MLP."""
    KIND = "nfm"

    def __init__(self, dataRecorder, lr, gpu, hidden_dim_list=(1024, 512, 256), hidden_dim=4, activation_func="ReLU",
                 normalization="bn", dropout_rate=0, regular=None, **kw):
        super().__init__("NFM", dataRecorder, lr, gpu, regular, **kw)
        nf, d = self.embedded_field_count(), self.dataRecorder.embedding_dim
        self.hidden_dim = hidden_dim
        self.v = nn.Parameter(torch.randn(nf * d, hidden_dim))
        self.linear = Linear(nf, 1)
        self.deep = _mlp_stack(hidden_dim, list(hidden_dim_list), activation_func, normalization, dropout_rate)
        self._finish()

    def logits(self, x):
        x = self.features_to_divice(x)
        idx = self.embedding_layer.index_matrix(x)
        first_order_output = self.linear(idx.t().float())
        emb_out = self.embedding_layer(x)
        second_order_output = X.nfm_bi_interaction(emb_out, self.v)
        return first_order_output + self.deep(second_order_output)


class AFM(_CTRModel):
    """model/AFM/AFM.py:19-45: LR + p(sum_pairs softmax(attention(e_i*e_j)) e_i*e_j)."""
    KIND = "afm"

    def __init__(self, dataRecorder, lr, gpu, attention_dim=8, regular=None, **kw):
        super().__init__("AFM", dataRecorder, lr, gpu, regular, **kw)
        self.attention_dim = attention_dim
        self.linear = LinearRegression_layer(self.dataRecorder)
        self.pool = X.AttentionPooling(self.dataRecorder.embedding_dim, attention_dim)
        self._finish()

    # the reference registers `attention` and `p` on the model itself: keep those names reachable
    @property
    def attention(self):
        return self.pool.attention

    @property
    def p(self):
        return self.pool.p

    def logits(self, x):
        x = self.features_to_divice(x)
        return self.linear(x) + self.pool(self.embedding_layer(x))


class DCN(_CTRModel):
    """model/DCN/DCN.py:16-52: cross network and MLP side by side on the flattened embeddings, then
    Linear(D + h_last, 1)."""
    KIND = "dcn"

    def __init__(self, dataRecorder, lr, gpu, hidden_dim_list=(1024, 512, 256), cross_layers_num=3, activation_func="ReLU",
                 normalization="bn", dropout_rate=0, regular=None, **kw):
        super().__init__("DCN", dataRecorder, lr, gpu, regular, **kw)
        D = self.embedded_field_count() * self.dataRecorder.embedding_dim
        self.cross = X.CrossNetwork(D, cross_layers_num)
        self.deep = _mlp_stack(D, list(hidden_dim_list), activation_func, normalization, dropout_rate, head=False)
        self.fc = Linear(D + hidden_dim_list[-1], 1)
        self._finish()

    def crossing_network(self, emb_out):
        return self.cross(emb_out)

    def logits(self, x):
        x = self.features_to_divice(x)
        emb_out = self.embedding_layer(x)
        cross_output = self.crossing_network(emb_out)
        deep_output = self.deep(emb_out)
        # Linear(D + h_last, 1) over the two blocks as they are (no concatenation, no split of its gradient)
        z = Fn.linear1_of_parts(cross_output, deep_output, self.fc.weight, self.fc.bias)
        return z if z is not None else self.fc(torch.cat([cross_output, deep_output], dim=1))
