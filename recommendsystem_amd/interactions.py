"""Feature-interaction operators of the reference's model zoo on HIP kernels (autograd wrappers).

    pnn_inner / pnn_outer       model/PNN/PNN.py:65-103
    CrossNetwork                model/DCN/DCN.py:16-17, 48-52
    nfm_bi_interaction          model/NFM/NFM.py:42-44
    AttentionPooling            model/AFM/AFM.py:19-26, 36-42, 47-56
    FieldAwareTables            model/FFM/FFM.py:21-23, 43-62

Forward and backward both run in librs_hip.so; nothing here falls back to torch arithmetic.
"""
from __future__ import annotations

import torch
from torch import nn

from . import functional as Fn
from . import ops
from ._lib import RSError
from .layers import _StackedTables


# -------------------------------------------------------------------------------------------------
# PNN
# -------------------------------------------------------------------------------------------------
class _PNNInnerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb):
        emb = emb.contiguous()
        ctx.save_for_backward(emb)
        return ops.pnn_inner_fwd(emb)

    @staticmethod
    def backward(ctx, g):
        (emb,) = ctx.saved_tensors
        return ops.pnn_inner_bwd(emb, g.contiguous())


class _PNNOuterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb):
        emb = emb.contiguous()
        ctx.save_for_backward(emb)
        return ops.pnn_outer_fwd(emb)

    @staticmethod
    def backward(ctx, g):
        (emb,) = ctx.saved_tensors
        return ops.pnn_outer_bwd(emb, g.contiguous())


def pnn_inner(emb: torch.Tensor) -> torch.Tensor:
    """emb (B, F, d) -> (B, F(F-1)/2): <e_i, e_j> for i<j in row-major pair order (PNN.py:65-77)."""
    return _PNNInnerFn.apply(emb)


def pnn_outer(emb: torch.Tensor) -> torch.Tensor:
    """emb (B, F, d) -> (B, F(F-1)/2 * d * d): e_i e_j^T flattened, i index major (PNN.py:80-103)."""
    return _PNNOuterFn.apply(emb)


# -------------------------------------------------------------------------------------------------
# DCN cross network
# -------------------------------------------------------------------------------------------------
def _end_to_end(ts) -> bool:
    """The tensors lie one behind the other in one buffer (a trainer's flat parameter / gradient buffer)."""
    p = ts[0].data_ptr()
    for t in ts:
        if not t.is_contiguous() or t.dtype != torch.float32 or t.data_ptr() != p:
            return False
        p += t.numel() * 4
    return True


def _rows_view(ts, rows: int, width: int) -> torch.Tensor:
    """[rows, width] over tensors that lie end to end (no copy), else their concatenation."""
    if _end_to_end(ts):
        return torch.as_strided(ts[0].detach().view(-1), (rows, width), (width, 1))
    return torch.cat([t.detach().reshape(1, -1) for t in ts], 0)


class _CrossFn(torch.autograd.Function):
    """The L cross layers in one launch each way.  The 2L parameters come in one by one: inside a trainer they are
    neighbouring views of its flat buffers, so w / b [L, D] are views (no concatenation launch) and the parameter
    gradients land in the flat gradient buffer directly (no per-parameter accumulation launch)."""

    @staticmethod
    def forward(ctx, x0, L, *params):
        x0 = x0.contiguous()
        D = x0.shape[1]
        w, b = _rows_view(params[:L], L, D), _rows_view(params[L:], L, D)
        out, S = ops.dcn_cross_fwd(x0, w, b)
        ctx.save_for_backward(x0, w, b, S)
        ctx.pids = [id(p) for p in params]
        ctx.shapes = [p.shape for p in params]
        return out

    @staticmethod
    def backward(ctx, g):
        x0, w, b, S = ctx.saved_tensors
        L, D = w.shape
        cap = Fn.TableCapture.active
        if cap is not None and cap.grad_sinks:
            peek = [cap.grad_sinks.get(pid) for pid in ctx.pids]
            if all(t is not None for t in peek) and not any(pid in cap._sunk for pid in ctx.pids) and _end_to_end(peek):
                for pid in ctx.pids:
                    cap.sink(pid)
                g_out = torch.as_strided(peek[0].view(-1), (2 * L * D,), (1,))
                dx0, _, _ = ops.dcn_cross_bwd(x0, S, g.contiguous(), w, b, g_out=g_out, parts=cap.colsum)
                return (dx0, None, *([None] * (2 * L)))
        dx0, dW, dB = ops.dcn_cross_bwd(x0, S, g.contiguous(), w, b)
        grads = [dW[l].view(ctx.shapes[l]) for l in range(L)] + [dB[l].view(ctx.shapes[L + l]) for l in range(L)]
        return (dx0, None, *grads)


class CrossNetwork(nn.Module):
    """x_{l+1} = x0 * (x_l @ w_l) + b_l + x_l on flattened (B, D) embeddings.

    Parameters as in DCN.py:16-17: ``cross_w[l]``, ``cross_b[l]`` of shape (D, 1) ~ randn.  The reference
    adds the (D, 1) bias to a (B, D) tensor, which only broadcasts when B == D; the bias is applied
    along the feature axis here (b_l^T), the evident intent."""

    def __init__(self, input_dim: int, cross_layers_num: int = 3):
        super().__init__()
        self.input_dim = input_dim
        self.cross_layers_num = cross_layers_num
        self.cross_w = nn.ParameterList([nn.Parameter(torch.randn(input_dim, 1)) for _ in range(cross_layers_num)])
        self.cross_b = nn.ParameterList([nn.Parameter(torch.randn(input_dim, 1)) for _ in range(cross_layers_num)])

    def forward(self, x0):
        x0 = x0.reshape(x0.shape[0], -1)
        return _CrossFn.apply(x0, self.cross_layers_num, *self.cross_w, *self.cross_b)


# -------------------------------------------------------------------------------------------------
# NFM (reference form)
# -------------------------------------------------------------------------------------------------
class _NFMBiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E, V):
        E, V = E.contiguous(), V.contiguous()
        out, s = ops.nfm_bi_fwd(E, V)
        ctx.save_for_backward(E, V, s)
        return out

    @staticmethod
    def backward(ctx, g):
        E, V, s = ctx.saved_tensors
        return ops.nfm_bi_bwd(E, V, s, g.contiguous())


def nfm_bi_interaction(emb_flat: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """0.5 * ((E v)^2 - E^2 v^2): (B, D) x (D, k) -> (B, k)   (NFM.py:42-44)."""
    return _NFMBiFn.apply(emb_flat.reshape(emb_flat.shape[0], -1), v)


# -------------------------------------------------------------------------------------------------
# AFM
# -------------------------------------------------------------------------------------------------
class _AFMFn(torch.autograd.Function):
    """Inside a trainer the five parameters are neighbouring views of its flat buffers, in the order the kernel
    leaves their gradients in: those then land in the flat gradient buffer directly (no per-parameter accumulation
    launch)."""

    @staticmethod
    def forward(ctx, emb, W, bA, h, pw, pb):
        emb = emb.contiguous()
        ctx.pids = [id(t) for t in (W, bA, h, pw, pb)]
        args = [t.contiguous() for t in (W, bA, h, pw, pb)]
        ctx.save_for_backward(emb, *args)
        ctx.shapes = (h.shape, pw.shape)
        return ops.afm_fwd(emb, *args).view(-1, 1)

    @staticmethod
    def backward(ctx, g):
        emb, W, bA, h, pw, pb = ctx.saved_tensors
        cap = Fn.TableCapture.active
        if cap is not None and cap.grad_sinks:
            peek = [cap.grad_sinks.get(pid) for pid in ctx.pids]
            if all(t is not None for t in peek) and not any(pid in cap._sunk for pid in ctx.pids) and _end_to_end(peek):
                for pid in ctx.pids:
                    cap.sink(pid)
                n = sum(t.numel() for t in peek)
                g_out = torch.as_strided(peek[0].view(-1), (n,), (1,))
                demb = ops.afm_bwd(emb, g.contiguous().view(-1), W, bA, h, pw, g_out=g_out, parts=cap.colsum)[0]
                return demb, None, None, None, None, None
        demb, dW, dbA, dh, dpw, dpb = ops.afm_bwd(emb, g.contiguous().view(-1), W, bA, h, pw)
        return demb, dW, dbA, dh.view(ctx.shapes[0]), dpw.view(ctx.shapes[1]), dpb


class AttentionPooling(nn.Module):
    """Pair-wise interaction layer + attention network + projection of AFM.

    Sub-modules keep the reference's names and shapes (AFM.py:19-26): ``attention`` =
    Sequential(Linear(d, A), ReLU, Linear(A, 1, bias=False), Softmax(dim=1)) and ``p`` = Linear(d, 1);
    they only hold the parameters -- the fused kernel does the arithmetic."""

    def __init__(self, emb_dim: int, attention_dim: int = 8):
        super().__init__()
        self.attention = nn.Sequential(nn.Linear(emb_dim, attention_dim), nn.ReLU(),
                                       nn.Linear(attention_dim, 1, bias=False), nn.Softmax(dim=1))
        self.p = nn.Linear(emb_dim, 1)

    def forward(self, emb):
        a0, a2 = self.attention[0], self.attention[2]
        return _AFMFn.apply(emb, a0.weight, a0.bias, a2.weight, self.p.weight, self.p.bias)


# -------------------------------------------------------------------------------------------------
# FFM
# -------------------------------------------------------------------------------------------------
class _FFMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, holder, *weights):
        T = holder.stacked_for(weights)
        holder.before_read(idx)
        ctx.holder = holder
        ctx.save_for_backward(idx, T)
        return ops.ffm_fwd(idx, holder.layout, T).view(-1, 1)

    @staticmethod
    def backward(ctx, g):
        idx, T = ctx.saved_tensors
        holder = ctx.holder
        layout = holder.layout
        plan = holder.plan_for(idx.shape[1], 1)            # only the sort/segment plan is used
        dT = ops.ffm_bwd(plan, idx, T, g.contiguous().view(-1))
        grads = [dT[o:o + v] for o, v in zip(layout.row_offsets, layout.vocab_sizes)]
        return (None, None, *grads)


class FieldAwareTables(nn.Module):
    """One latent vector per (feature value, partner field): ``tables[col]`` is an
    nn.Embedding(V_col, F*d, padding_idx=0) whose row holds the F vectors of that value.

    The reference builds ``embedding_dict[feature] = [layer for _ in range(field_num)]`` (FFM.py:22-23),
    F aliases of one table, which collapses FFM into FM; independent tables are what
    ``ffm_interaction`` (FFM.py:43-62) is written for and what this module provides.  Initialisation
    follows Layers.py:130-135 (rows 1.. ~ N(0, 1e-4^2), row 0 zero)."""

    def __init__(self, feature_map, columns, emb_dim: int, init_std: float = 1e-4):
        super().__init__()
        self.columns = list(columns)
        self.emb_dim = emb_dim
        self.field_num = len(self.columns)
        mods = {}
        for c in self.columns:
            e = nn.Embedding(feature_map[c]["vocab_size"], self.field_num * emb_dim, padding_idx=0)
            nn.init.normal_(e.weight[1:, :], mean=0.0, std=init_std)
            mods[c] = e
        self.tables = nn.ModuleDict(mods)
        self._tab = _StackedTables(self.tables, self.columns, self.field_num * emb_dim)
        self._tab.restack()

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self._tab.restack()
        return r

    def forward(self, idx):
        """idx: columnar [F, B] int64 on the device -> (B, 1) second-order term."""
        if not idx.is_cuda:
            raise RSError("FieldAwareTables: the batch must be on the device")
        from .functional import TableCapture
        cap = TableCapture.active
        if cap is not None and cap.provider is not None:
            G = cap.provider.ffm_rows(idx, self._tab)          # [B, F, F*d] through the row-sharded exchange
            out = cap.record("ffm", self._tab, idx, ops.ffm_rows_fwd(G, self.emb_dim))
            cap.records[-1]["rows"] = G
            return out.view(-1, 1)
        if cap is not None:
            T = self._tab.stacked_for(self._tab.weights())
            cap.before_lookup(self._tab, idx)
            return cap.record("ffm", self._tab, idx, ops.ffm_fwd(idx, self._tab.layout, T)).view(-1, 1)
        return _FFMFn.apply(idx, self._tab, *self._tab.weights())
