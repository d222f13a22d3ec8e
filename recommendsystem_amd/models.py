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
from .layers import LinearRegression_layer, MLPBlock, ResidualMLPBlock


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
    return nn.Sequential(*layers)


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
