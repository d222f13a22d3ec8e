"""Autograd bridges between the reference-shaped nn.Modules and the HIP kernels.

Forward and backward of every op run in librs_hip.so; torch is used for tensor allocation,
views and autograd bookkeeping only.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from .ops import ACT_CODES, FieldLayout, SparsePlan

_FUSABLE = (0, 1, 2, 3)      # none / relu / tanh / leakyrelu can ride in the GEMM epilogue
_SIGN_ACTS = (0, 1, 3)       # none / relu / leakyrelu: the derivative is a function of the sign of the output
_ONE_WIDE_MAX_K = 1024       # widest input rs_head_bwd takes
_PAD_MIN_K = 64              # narrower odd inputs are not worth the two padding copies


def stack_columns(x, cols) -> torch.Tensor:
    """dict of per-column (B,) tensors -> columnar [F, B] int64 index matrix (one copy)."""
    ts = [x[c] for c in cols]
    if all(t.dtype == torch.int64 for t in ts):
        # columns that are the rows of one [F, B] matrix already (a columnar batch handed over as a dict of views):
        # the matrix itself, no copy
        t0 = ts[0]
        if t0.dim() == 1 and t0.is_contiguous():
            B, base, off0 = t0.shape[0], t0.untyped_storage().data_ptr(), t0.storage_offset()
            if all(t.dim() == 1 and t.shape[0] == B and t.is_contiguous() and t.untyped_storage().data_ptr() == base
                   and t.storage_offset() == off0 + i * B for i, t in enumerate(ts)):
                return t0.as_strided((len(ts), B), (B, 1))
        return torch.stack(ts, 0)
    return torch.stack([t.long() for t in ts], 0)      # Layers.py:147 `.long()`


# -------------------------------------------------------------------------------------------------
# embedding lookup (d-dim and 1-dim) with dense per-column gradients
# -------------------------------------------------------------------------------------------------
class _GatherFn(torch.autograd.Function):
    """out[b, f, :] = table[row_offset[f] + idx[f, b], :] for the stacked table behind `weights`.

    `weights` are the per-column nn.Embedding weights, all views into `stacked`; the backward
    returns per-column views into ONE dense gradient buffer filled by the sort + segmented
    reduce + scatter kernels (deterministic; padding row 0 never receives gradient).
    """

    @staticmethod
    def forward(ctx, idx, holder, *weights):
        stacked = holder.stacked_for(weights)
        holder.before_read(idx)
        out = ops.gather_fwd(idx, holder.layout, stacked)
        ctx.holder = holder
        ctx.save_for_backward(idx)
        ctx.d = stacked.shape[1]
        return out

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        holder = ctx.holder
        layout: FieldLayout = holder.layout
        Fn, B = idx.shape
        d = ctx.d
        plan = holder.plan_for(B, d)
        plan.sort_columnar(idx)
        partials = ops.default_workspace(idx.device).get("gather_bwd_partials", plan.n_partials)
        plan.reduce(partials, gout=gout.contiguous().view(B * Fn, d))
        grad = torch.zeros((layout.total_rows, d), dtype=torch.float32, device=idx.device)
        plan.scatter_dense(grad=grad)
        grads = [grad[o:o + v] for o, v in zip(layout.row_offsets, layout.vocab_sizes)]
        return (None, None, *grads)


class TableCapture:
    """``with TableCapture() as cap:`` -- inside, the table lookups of this package (embedding rows, the one-hot
    LR sum, the FFM term) do not build dense table gradients.  Each returns a leaf tensor and is recorded as
    dict(kind, holder, idx, out); after ``loss.backward()`` the ``out.grad`` of a record is exactly what the
    touched-rows kernels consume (engine.SparseTrainer).  Dense parameters keep their normal autograd path."""

    active = None

    def __init__(self, provider=None, grad_sinks=None, splitk=None, on_lookup=None, colsum=None):
        # provider: object with emb(idx, holder) -> [B, F, d] and lr_terms(idx, holder) -> [B, F] (the per-field
        # 1-dim values) that fetches rows from somewhere else than the local stacked table (row-sharded tables)
        # grad_sinks: id(parameter) -> zero-initialised gradient tensor the MLP-block backward writes into directly
        # (instead of returning a gradient for autograd to add); splitk: an ops.SplitKParts that collects the
        # split-K slabs of those weight gradients -- the trainer calls its finish() once after backward()
        self.records = []
        self.provider = provider
        self.grad_sinks = grad_sinks or {}
        self.splitk = splitk
        # colsum: an ops.ColsumParts shared by the whole backward pass: bias gradients (and other column sums) whose
        # destination is a grad sink leave their partial sums there and the trainer finishes them all in one launch
        self.colsum = colsum
        # on_lookup(holder, idx): called before a local table is read (the trainer brings the rows up to date)
        self.on_lookup = on_lookup
        self.rng_step = None     # device int64 counter that numbers the dropout masks of this pass (the trainer's step count)
        self._sunk = set()
        self._idx = {}           # (id of the batch dict, columns) -> (batch dict, index matrix): one stack per forward

    def index_matrix(self, x, cols, build):
        """The columnar index matrix of batch dict `x` for `cols`, built once per forward pass: lookups of the
        same batch then share one tensor (and the trainer one sort)."""
        key = (id(x), tuple(cols))
        hit = self._idx.get(key)
        if hit is None:
            hit = self._idx[key] = (x, build())
        return hit[1]

    def __enter__(self):
        if TableCapture.active is not None:
            raise RuntimeError("TableCapture is not re-entrant")
        TableCapture.active = self
        return self

    def __exit__(self, *exc):
        TableCapture.active = None
        return False

    def sink(self, pid):
        """The gradient destination of parameter `pid` on its first use in this pass (a parameter that appears twice
        in the graph gets its further contributions through autograd's accumulation)."""
        t = self.grad_sinks.get(pid)
        if t is None or pid in self._sunk:
            return None
        self._sunk.add(pid)
        return t

    def before_lookup(self, holder, idx):
        if self.on_lookup is not None:
            self.on_lookup(holder, idx)
        else:
            holder.before_read(idx)

    def record(self, kind, holder, idx, out):
        out.requires_grad_(True)
        self.records.append({"kind": kind, "holder": holder, "idx": idx, "out": out})
        return out


def embedding_lookup(idx, holder, weights) -> torch.Tensor:
    cap = TableCapture.active
    if cap is not None:
        if cap.provider is not None:
            return cap.record("emb", holder, idx, cap.provider.emb(idx, holder))
        stacked = holder.stacked_for(weights)
        cap.before_lookup(holder, idx)
        return cap.record("emb", holder, idx, ops.gather_fwd(idx, holder.layout, stacked))
    return _GatherFn.apply(idx, holder, *weights)


class _HeadSum:
    """sum_b dL/dlogit[b] of the last loss head, which its launch leaves behind for nothing (bce_of_logits): whoever
    needs the batch sum of exactly that gradient -- the bias of the LR layer (Layers.py:104-109), the bias of a 1-wide
    last Linear (DCN.py:31) -- takes it from here instead of summing B values in two launches.  `dz` is kept alive so
    that no other tensor can sit at its address while the entry is valid."""
    dz = None
    total = None


def batch_sum_of(g, out=None, parts=None):
    """sum over the batch of the gradient g ([B] or [B, 1]) as a 1-element tensor (written to `out` when given).  parts:
    a ColsumParts collector whose finish() the caller runs anyway; the result is then valid after it."""
    hs = _HeadSum
    if (hs.dz is not None and g.is_contiguous() and g.numel() == hs.dz.numel() and g.data_ptr() == hs.dz.data_ptr()):
        if out is None:
            return hs.total
        if parts is not None:
            parts.add(hs.total, 1, 1, 1, dest=out)
            return out
        out.copy_(hs.total)
        return out
    return ops.colsum(g.reshape(-1, 1), out=out)


class _LRFn(torch.autograd.Function):
    """(B, 1) = sum_f w_f[idx[f, b]] + bias   (Layers.py:107-109)."""

    @staticmethod
    def forward(ctx, idx, holder, bias, *weights1):
        stacked1 = holder.stacked_for(weights1)
        holder.before_read(idx)
        out = ops.gather_lr_fm_fwd(idx, holder.layout, None, stacked1, bias, want_emb=False, want_lr=True)["lr"]
        ctx.holder = holder
        ctx.save_for_backward(idx)
        return out.view(-1, 1)

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        holder = ctx.holder
        layout = holder.layout
        Fn, B = idx.shape
        g = gout.contiguous().view(B)
        plan = holder.plan_for(B, 1, with_lr=True)
        plan.sort_columnar(idx)
        partials = ops.default_workspace(idx.device).get("gather_bwd_partials", plan.n_partials)
        plan.reduce(partials, dlr=g)
        grad1 = torch.zeros((layout.total_rows, 1), dtype=torch.float32, device=idx.device)
        plan.scatter_dense(grad1=grad1)
        grads = [grad1[o:o + v] for o, v in zip(layout.row_offsets, layout.vocab_sizes)]
        dbias = ops.colsum(g.view(B, 1))
        return (None, None, dbias, *grads)


class _BiasGradFn(torch.autograd.Function):
    """Identity on x (which already contains the bias); routes the column sum of the gradient to the bias."""

    @staticmethod
    def forward(ctx, x, bias):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return g, batch_sum_of(g)


def lr_forward(idx, holder, bias, weights1) -> torch.Tensor:
    cap = TableCapture.active
    if cap is not None:
        if cap.provider is not None:
            terms = cap.provider.lr_terms(idx, holder)            # [B, F]
            ones = torch.ones(terms.shape[1], dtype=torch.float32, device=terms.device)
            raw = ops.rowdot_fwd(terms, ones, bias.detach(), 0)     # sum_f w_f[idx] + bias
        else:
            stacked1 = holder.stacked_for(weights1)
            cap.before_lookup(holder, idx)
            raw = ops.gather_lr_fm_fwd(idx, holder.layout, None, stacked1, bias.detach(), want_emb=False,
                                       want_lr=True)["lr"]
        return _BiasGradFn.apply(cap.record("lr", holder, idx, raw).view(-1, 1), bias)
    return _LRFn.apply(idx, holder, bias, *weights1)


# -------------------------------------------------------------------------------------------------
# MLP blocks
# -------------------------------------------------------------------------------------------------
class MLPCtx:
    __slots__ = ("x", "W", "h", "z", "y", "xhat", "var", "rstd", "act", "norm", "gamma", "slope", "eps", "k", "one",
                 "linear", "drop_fused", "pre")


def _pad_columns(t, pad: int):
    """[M, K] -> [M, K + pad] with zero columns behind (one copy and a fill of the pad columns only)."""
    out = t.new_empty((t.shape[0], t.shape[1] + pad))
    out[:, :t.shape[1]].copy_(t)
    out[:, t.shape[1]:].zero_()
    return out


def mlp_block_fwd(x2d, W, b, act: int, norm: Optional[str], gamma=None, beta=None, slope=None, training=True,
                  running_mean=None, running_var=None, eps=1e-5, allow_pad=False, linear=None, dropout=None):
    """act(norm(x W^T + b)) for one block (Layers.py:39-47).  Returns (h, ctx).

    allow_pad: the caller's backward is mlp_block_bwd (which knows about a padded input width).  x2d may come in
    padded already (zero columns up to the next multiple of four behind W's input width: pnn_inner_input); its
    gradient then keeps that width.
    linear: an object whose forward() returns x W^T + b and whose backward(dz) returns dict(dx, dW, db) in place of the
    plain GEMMs (OuterLinear: an input too wide to exist at once); x2d is then ignored.
    dropout: the (p, seed, salt, call_no, call_dev) key of the block's dropout.  Where the mask can ride in the GEMM's
    epilogue (no normalisation, an activation whose derivative only needs the sign of its output) the returned h is
    already masked and ctx.drop_fused is set; otherwise the caller applies the mask."""
    c = MLPCtx()
    c.x, c.W, c.act, c.norm, c.gamma, c.slope, c.eps = x2d, W, act, norm, gamma, slope, eps
    c.z = c.y = c.xhat = c.var = c.rstd = None
    c.k = W.shape[1]
    c.linear = linear
    c.drop_fused = False
    c.pre = linear is None and x2d.shape[1] != c.k
    if c.pre and not (allow_pad and x2d.shape[1] == (c.k + 3) // 4 * 4):
        raise ops.RSError(f"mlp_block: input has {x2d.shape[1]} columns, the weight {c.k}")
    if linear is not None:
        c.one = False
        z = linear.forward()
        fus = act if act in _FUSABLE else 0
        if not norm:
            if act in _FUSABLE:
                c.h = ops.act_fwd(z, act) if act else z
            else:
                c.z = z
                c.h = ops.act_fwd(z, act, slope)
            return c.h, c
        if norm == "bn":
            y, c.xhat, c.var = ops.bn_fwd(z, gamma, beta, fus, training, running_mean, running_var, eps)
        elif norm == "ln":
            y, c.xhat, c.rstd = ops.ln_fwd(z, gamma, beta, fus, eps)
        else:
            raise ValueError("normalization must be either 'bn' or 'ln' or None")
        if act in _FUSABLE:
            c.h = y
        else:
            c.y = y
            c.h = ops.act_fwd(y, act, slope)
        return c.h, c
    # a 1-wide plain layer is a dot product per row, not a GEMM
    c.one = (W.shape[0] == 1 and not norm and act in _FUSABLE and b is not None and W.shape[1] <= _ONE_WIDE_MAX_K
             and not c.pre and (W.shape[1] % 4 != 0 or (x2d.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0)))
    if c.one:
        c.h = ops.rowdot_fwd(x2d, W.reshape(-1), b, act).view(-1, 1)
        return c.h, c
    if c.pre:
        c.W = W = _pad_columns(W.detach(), x2d.shape[1] - c.k)
    elif allow_pad and c.k % 4 != 0 and c.k >= _PAD_MIN_K and x2d.shape[0] >= 256:
        # rows of an odd width are not 16-byte aligned and would take the register-staged GEMM; zero columns up to
        # the next multiple of four change no sum and let all three products of the block run on the DMA kernel
        c.x = x2d = _pad_columns(x2d, 4 - c.k % 4)
        c.W = W = _pad_columns(W.detach(), 4 - c.k % 4)
    fus = act if act in _FUSABLE else 0
    if not norm:
        if act in _FUSABLE:
            # relu / leakyrelu / none: act'(h) and act'(mask * h) agree wherever the mask lets a gradient through
            c.drop_fused = dropout is not None and dropout[0] > 0 and act in _SIGN_ACTS
            c.h = ops.gemm(x2d, W, bias=b, act=act, dropout=dropout if c.drop_fused else None)
        else:                                   # PReLU: keep z for the slope gradient
            c.z = ops.gemm(x2d, W, bias=b)
            c.h = ops.act_fwd(c.z, act, slope)
        return c.h, c
    z = ops.gemm(x2d, W, bias=b)
    if norm == "bn":
        y, c.xhat, c.var = ops.bn_fwd(z, gamma, beta, fus, training, running_mean, running_var, eps)
    elif norm == "ln":
        y, c.xhat, c.rstd = ops.ln_fwd(z, gamma, beta, fus, eps)
    else:
        raise ValueError("normalization must be either 'bn' or 'ln' or None")
    if act in _FUSABLE:
        c.h = y
    else:
        c.y = y
        c.h = ops.act_fwd(y, act, slope)
    return c.h, c


def mlp_block_bwd(dh, c: MLPCtx, need_dx=True, dz_ready=False, dW_out=None, db_out=None, mulsrc=None, mul_act=0,
                  defer_reduce=None, colsum_parts=None, dbp_out=None, db_done=False, dx_dropout=None, db_parts=None):
    """Backward of one block.  Returns dict(dx, dW, db, dgamma, dbeta, dslope).

    dz_ready: `dh` already is dL/dz (the caller fused act' into the GEMM that produced it).
    mulsrc/mul_act: fuse the PREVIOUS block's act' into this block's dx GEMM, so that the returned
    dx is the previous block's dL/dz.
    colsum_parts/dbp_out: with a fused act', the launch that produces dx also leaves the partial column sums of dx --
    the previous block's bias gradient -- with the collector (valid in dbp_out after its finish()).
    db_done: this block's bias gradient is already on its way (the block above left it with the collector).
    db_parts: a collector the caller finishes later; a bias gradient that needs a column-sum pass of its own then only
    leaves its partial sums there (db_out must be given and must not be read before that finish()).
    dx_dropout: the previous block's dropout key, to be multiplied into dx in the same epilogue (with mulsrc)."""
    out = {"dgamma": None, "dbeta": None, "dslope": None, "db": None}
    fus = c.act if c.act in _FUSABLE else 0
    if not c.norm:
        if dz_ready:
            dz = dh
        elif c.act in _FUSABLE:
            dz = ops.act_bwd(dh, c.act, h=c.h)[0] if c.act != 0 else dh
        else:
            dz, out["dslope"] = ops.act_bwd(dh, c.act, z=c.z, slope=c.slope)
    else:
        if c.act not in _FUSABLE:
            dh, out["dslope"] = ops.act_bwd(dh, c.act, z=c.y, slope=c.slope)
            hh = c.y
        else:
            hh = c.h
        if c.norm == "bn":
            dz, out["dgamma"], out["dbeta"] = ops.bn_bwd(dh, hh, c.xhat, fus, c.gamma, c.var, c.eps)
        else:
            dz, out["dgamma"], out["dbeta"] = ops.ln_bwd(dh, hh, c.xhat, c.rstd, c.gamma, fus)
    if getattr(c, "linear", None) is not None:
        r = c.linear.backward(dz, dW_out=dW_out, db_out=db_out)
        out.update(r)
        out["dz"] = dz
        return out
    if getattr(c, "one", False):
        # 1-wide layer: dx, the partial sums of dW and (with a fused act') of the previous block's bias gradient
        # come out of one launch
        own = colsum_parts is None
        parts = colsum_parts if colsum_parts is not None else ops.ColsumParts(ops.default_workspace(dz.device))
        dW = dW_out if dW_out is not None else torch.empty_like(c.W)
        if not db_done:
            out["db"] = batch_sum_of(dz, out=db_out, parts=parts)
        out["dx"] = ops.head_bwd(dz.reshape(-1), c.W.reshape(-1), c.x, mul_act if mulsrc is not None else 0, parts,
                                 dW_out=dW.view(-1), dbp_out=dbp_out if mulsrc is not None else None)
        if own:
            parts.finish()
        out["dbp_left"] = mulsrc is not None and dbp_out is not None
        out["dW"] = dW
        out["dz"] = dz
        return out
    if c.x.shape[1] != c.k:
        # padded input width (mlp_block_fwd): the pad columns of dW and dx are zero and are cut off again
        dxp = None
        if ops.GROUP_STAGES and need_dx:
            # both products need only dz: one launch, the padded dW's slabs reduced right after it
            grp = ops.GemmGroup(ops.GROUP_STAGES)
            skp = ops.SplitKParts(ops.default_workspace(dz.device), tag="pad")
            dWp = grp.add(dz, c.x, a_mc=True, b_nc=True, defer_reduce=skp)
            dxp = grp.add(dz, c.W, b_nc=True)
            grp.launch()
            skp.finish()
        else:
            dWp = ops.gemm(dz, c.x, a_mc=True, b_nc=True)
        if dW_out is not None:
            dW_out.copy_(dWp[:, :c.k])
            out["dW"] = dW_out
        else:
            out["dW"] = dWp[:, :c.k]
        if not db_done:
            out["db"] = ops.colsum(dz, out=db_out, parts=db_parts if db_out is not None else None)
        out["dx"] = None
        if need_dx:
            out["dx"] = dxp if dxp is not None else ops.gemm(dz, c.W, b_nc=True)
            if not c.pre:
                out["dx"] = out["dx"][:, :c.k]
        if need_dx and mulsrc is not None:
            out["dx"] = ops.act_bwd(out["dx"].contiguous(), mul_act, h=mulsrc)[0]
        out["dz"] = dz
        return out
    # dW and dx both need only dz: one launch for the two products (ops.GemmGroup)
    grp = ops.GemmGroup(ops.GROUP_STAGES) if (ops.GROUP_STAGES and need_dx) else None
    if grp is not None:
        out["dW"] = grp.add(dz, c.x, a_mc=True, b_nc=True, out=dW_out, defer_reduce=defer_reduce)
    else:
        out["dW"] = ops.gemm(dz, c.x, a_mc=True, b_nc=True, out=dW_out, defer_reduce=defer_reduce)
    if not db_done:
        out["db"] = ops.colsum(dz, out=db_out, parts=db_parts if db_out is not None else None)
    if need_dx:
        if mulsrc is not None and colsum_parts is not None and dbp_out is not None:
            colsum_parts.expect(dbp_out)
            if grp is not None:
                out["dx"] = grp.add(dz, c.W, b_nc=True, mulsrc=mulsrc, mul_act=mul_act, colsum_part=colsum_parts,
                                    dropout=dx_dropout)
            else:
                out["dx"] = ops.gemm(dz, c.W, b_nc=True, mulsrc=mulsrc, mul_act=mul_act, colsum_part=colsum_parts,
                                     dropout=dx_dropout)
            out["dbp_left"] = True
        elif grp is not None:
            out["dx"] = grp.add(dz, c.W, b_nc=True, mulsrc=mulsrc, mul_act=mul_act, dropout=dx_dropout)
        else:
            out["dx"] = ops.gemm(dz, c.W, b_nc=True, mulsrc=mulsrc, mul_act=mul_act, dropout=dx_dropout)
        if grp is not None:
            grp.launch()
    else:
        out["dx"] = None
    out["dz"] = dz
    return out


class _MLPBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, W, b, gamma, beta, slope, act, norm, training, running_mean, running_var, eps):
        h, c = mlp_block_fwd(x2d, W, b, act, norm, gamma, beta, slope, training, running_mean, running_var, eps,
                             allow_pad=True)
        ctx.c = c
        ctx.has = (gamma is not None, slope is not None)
        ctx.pids = (id(W), id(b) if b is not None else None)
        # the output goes through save_for_backward: kept in ctx.c it would close a reference cycle (output ->
        # grad_fn -> ctx -> output) that only the garbage collector frees, at a time of its choosing
        c.h = None
        ctx.save_for_backward(h)
        return h

    @staticmethod
    def backward(ctx, dh):
        cap = TableCapture.active
        sW = sb = defer = None
        if cap is not None and cap.grad_sinks:
            sW = cap.sink(ctx.pids[0])
            sb = cap.sink(ctx.pids[1]) if ctx.pids[1] is not None else None
            defer = cap.splitk if sW is not None else None
        ctx.c.h, = ctx.saved_tensors
        # with both gradients going to their sinks nothing reads them before the trainer's one finishing launch
        shared = cap.colsum if (cap is not None and sW is not None and (sb is not None or ctx.pids[1] is None)) else None
        r = mlp_block_bwd(dh.contiguous(), ctx.c, need_dx=ctx.needs_input_grad[0], dW_out=sW, db_out=sb, defer_reduce=defer,
                          colsum_parts=shared if ctx.c.one else None, db_parts=shared)
        ctx.c.h = None
        # gradients written into their sinks are not handed to autograd a second time
        return (r["dx"], None if sW is not None else r["dW"], None if sb is not None else r["db"], r["dgamma"], r["dbeta"],
                r["dslope"], None, None, None, None, None, None)


class _Linear1Of2Fn(torch.autograd.Function):
    """nn.Linear(K1 + K2, 1) over cat([x1, x2], 1) (model/DCN/DCN.py:31,52) without the concatenation and without the
    copies that split its gradient again: one dot-product launch forward, one launch backward for dx1, dx2 and the
    partial sums of dW."""

    @staticmethod
    def forward(ctx, x1, x2, W, b):
        x1, x2 = x1.contiguous(), x2.contiguous()
        ctx.save_for_backward(x1, x2, W)
        ctx.pids = (id(W), id(b))
        return ops.rowdot2_fwd(x1, x2, W.detach().reshape(-1), b.detach(), 0).view(-1, 1)

    @staticmethod
    def backward(ctx, g):
        x1, x2, W = ctx.saved_tensors
        cap = TableCapture.active
        sW = sb = None
        if cap is not None and cap.grad_sinks:
            sW, sb = cap.sink(ctx.pids[0]), cap.sink(ctx.pids[1])
        g = g.contiguous()
        shared = cap.colsum if (cap is not None and sW is not None and sb is not None) else None
        parts = shared if shared is not None else ops.ColsumParts(ops.default_workspace(g.device))
        dW = sW if sW is not None else torch.empty_like(W)
        db = batch_sum_of(g, out=sb, parts=parts)
        dx1, dx2 = ops.head2_bwd(g.view(-1), W.detach().reshape(-1), x1, x2, 0, parts, dW_out=dW.view(-1))
        if shared is None:
            parts.finish()
        return dx1, dx2, None if sW is not None else dW, None if sb is not None else db


def linear1_of_parts(x1, x2, W, b):
    """[B, 1] = cat([x1, x2], 1) W^T + b for a 1-wide Linear, or None when the shape is not served."""
    if (W.shape[0] != 1 or b is None or x1.dim() != 2 or x2.dim() != 2 or not x1.is_cuda or W.shape[1] != x1.shape[1] + x2.shape[1]
            or W.shape[1] > _ONE_WIDE_MAX_K or x1.dtype != torch.float32 or x2.dtype != torch.float32):
        return None
    if (x1.shape[1] | x2.shape[1]) % 4 == 0 and any(t.data_ptr() % 16 != 0 or not t.is_contiguous() for t in (x1, x2, W)):
        return None                      # the vector loads of the kernel need 16-byte aligned rows
    return _Linear1Of2Fn.apply(x1, x2, W, b)


def mlp_block(x2d, W, b, act, norm, gamma, beta, slope, training, running_mean, running_var, eps=1e-5):
    return _MLPBlockFn.apply(x2d, W, b, gamma, beta, slope, act, norm, training, running_mean, running_var, eps)


# -------------------------------------------------------------------------------------------------
# PNN: the first MLP layer over [embeddings | inner products | outer products of all field pairs]
# -------------------------------------------------------------------------------------------------
OUTER_BLOCK_BYTES = 0                # size aimed at for one block of outer products; 0: the library's 192 MiB


class OuterLinear:
    """z = [x_other | outer(emb)] W^T + b without the (B, P*d*d) matrix of outer products (model/PNN/PNN.py:80-103
    feeding PNN.py:19-27,58: 189,696 columns at F = 39, d = 16, 3.1 GB at B = 4096).  The library walks the reduction
    dimension a block of pairs at a time (rs_pnn_outer_mlp_fwd / _bwd: products of the block written to a cache-sized
    buffer, multiplied into the running sum, buffer reused; the backward recomputes each block to form its columns
    of dW and its share of demb)."""

    def __init__(self, emb, x_other, W, b):
        self.emb = emb.contiguous()                       # [B, F, d]
        B, Fn, d = self.emb.shape
        K0 = 0 if x_other is None else x_other.shape[1]
        KO = Fn * (Fn - 1) // 2 * d * d
        if W.shape[1] != K0 + KO:
            raise ops.RSError(f"OuterLinear: weight has {W.shape[1]} columns, expected {K0} + {KO}")
        if x_other is not None and x_other.stride(-1) != 1:
            x_other = x_other.contiguous()
        self.x_other, self.W, self.b = x_other, W, b

    def forward(self):
        return ops.pnn_outer_mlp_fwd(self.emb, self.x_other, self.W.detach().contiguous(), self.b.detach(),
                                     block_bytes=OUTER_BLOCK_BYTES)

    def backward(self, dz, dW_out=None, db_out=None):
        dW, db, demb, dxo = ops.pnn_outer_mlp_bwd(self.emb, self.x_other, dz.contiguous(), self.W.detach().contiguous(),
                                                  dW_out=dW_out, db_out=db_out, block_bytes=OUTER_BLOCK_BYTES)
        return {"dx": None, "dW": dW, "db": db, "demb": demb, "dx_other": dxo}


class _PNNOuterBlockFn(torch.autograd.Function):
    """MLPBlock (Layers.py:39-47) whose input is [x_other | outer products of emb] (PNN.py:58 with pnn_type outer / both),
    evaluated through OuterLinear."""

    @staticmethod
    def forward(ctx, emb, x_other, W, b, gamma, beta, slope, act, norm, training, running_mean, running_var, eps):
        lin = OuterLinear(emb, x_other, W, b)
        h, c = mlp_block_fwd(None, W, b, act, norm, gamma, beta, slope, training, running_mean, running_var, eps, linear=lin)
        ctx.c = c
        ctx.pids = (id(W), id(b))
        c.h = None
        ctx.save_for_backward(h)
        ctx.has_other = x_other is not None
        return h

    @staticmethod
    def backward(ctx, dh):
        cap = TableCapture.active
        sW = sb = None
        if cap is not None and cap.grad_sinks:
            sW, sb = cap.sink(ctx.pids[0]), cap.sink(ctx.pids[1])
        ctx.c.h, = ctx.saved_tensors
        r = mlp_block_bwd(dh.contiguous(), ctx.c, dW_out=sW, db_out=sb)
        ctx.c.h = None
        return (r["demb"], r["dx_other"] if ctx.has_other else None, None if sW is not None else r["dW"],
                None if sb is not None else r["db"], r["dgamma"], r["dbeta"], r["dslope"], None, None, None, None, None, None)


def pnn_outer_block(block, emb, x_other):
    """`block` (an MLPBlock) applied to cat([x_other, outer products of emb], 1) without building that matrix."""
    gamma, beta, rm, rv, eps = block._norm_args()
    slope = block.activation_func.weight if block._act == ACT_CODES["prelu"] else None
    h = _PNNOuterBlockFn.apply(emb, x_other, block.fc.weight, block.fc.bias, gamma, beta, slope, block._act,
                               block.normalization or None, block.training, rm, rv, eps)
    return block.dropout(h)


class _PNNInnerInputFn(torch.autograd.Function):
    """cat([emb.flatten(1), inner products of all field pairs], 1) (model/PNN/PNN.py:58 with pnn_type inner, the
    products of PNN.py:65-77), zero columns up to the next multiple of four behind it: written by the product kernel in
    one launch, and its backward returns the embeddings' gradient through both paths at once."""

    @staticmethod
    def forward(ctx, emb):
        emb = emb.contiguous()
        ctx.save_for_backward(emb)
        return ops.pnn_inner_cat_fwd(emb)

    @staticmethod
    def backward(ctx, g):
        emb, = ctx.saved_tensors
        if g.stride(-1) != 1:
            g = g.contiguous()
        return ops.pnn_inner_cat_bwd(emb, g)


def pnn_inner_input(emb):
    """The (padded) input of PNN's first MLP layer for pnn_type inner, or None when the shape is not served by the
    concatenating kernel (the caller then concatenates itself)."""
    if emb.dim() != 3 or not emb.is_cuda or emb.dtype != torch.float32 or not ops.pnn_inner_cat_ok(emb.shape[1], emb.shape[2]):
        return None
    return _PNNInnerInputFn.apply(emb)


_dropout_calls = 0


class BlockSpec:
    """What _MLPStackFn needs to know about one block besides its parameters."""
    __slots__ = ("act", "norm", "training", "rm", "rv", "eps", "drop")

    def __init__(self, act, norm, training, rm, rv, eps, drop):
        self.act, self.norm, self.training, self.rm, self.rv, self.eps = act, norm, training, rm, rv, eps
        self.drop = drop          # None or the (p, seed, salt, call_no, call_dev) key of the block's dropout mask


class _MLPStackFn(torch.autograd.Function):
    """A run of MLPBlocks (Layers.py:8-47) as one autograd node, so that the backward pass can do what the
    block-by-block form cannot: a block's dx GEMM applies the previous block's act' in its epilogue and leaves the
    partial column sums of the result (the previous block's bias gradient) behind, one launch finishes all of them,
    and a 1-wide last block is a row-dot kernel whose backward emits dx and its weight gradient together."""

    @staticmethod
    def forward(ctx, x2d, specs, *params):
        cs = []
        h = x2d
        for i, s in enumerate(specs):
            W, b, gamma, beta, slope = params[5 * i:5 * i + 5]
            h, c = mlp_block_fwd(h, W, b, s.act, s.norm, gamma, beta, slope, s.training, s.rm, s.rv, s.eps, allow_pad=True,
                                 dropout=s.drop)
            cs.append(c)
            if s.drop is not None and not c.drop_fused:
                h = ops.dropout_apply(h, *s.drop)
        ctx.cs, ctx.specs = cs, specs
        ctx.pids = [id(p) if p is not None else None for p in params]
        ctx.shapes = [(p.shape if p is not None else None) for p in params]
        # the output goes through save_for_backward (kept in ctx.cs it would close a reference cycle output -> grad_fn
        # -> ctx -> output that only the garbage collector frees)
        ctx.out_is_h = h is cs[-1].h
        if ctx.out_is_h:
            cs[-1].h = None
            ctx.save_for_backward(h)
        return h

    @staticmethod
    def backward(ctx, dout):
        cs, specs = ctx.cs, ctx.specs
        if ctx.out_is_h:
            cs[-1].h, = ctx.saved_tensors
        n = len(specs)
        dev = dout.device
        cap = TableCapture.active
        sinks = cap is not None and bool(cap.grad_sinks)
        grads = [None] * (5 * n)
        dest = [None] * (5 * n)          # where the W / b gradient of a block is written
        sunk = [False] * (5 * n)
        for i in range(n):
            for j in (0, 1):
                k = 5 * i + j
                if ctx.pids[k] is None:
                    continue
                t = cap.sink(ctx.pids[k]) if sinks else None
                sunk[k] = t is not None
                dest[k] = t if t is not None else torch.empty(ctx.shapes[k], dtype=torch.float32, device=dev)
        # every gradient of the stack goes to a sink: nothing reads them before the trainer's one finishing launch
        shared = cap.colsum if (sinks and all(sunk[k] for k in range(5 * n) if dest[k] is not None)) else None
        parts = shared if shared is not None else ops.ColsumParts(ops.default_workspace(dev))
        dh = dout.contiguous()
        dz_ready = db_done = masked = False
        for i in range(n - 1, -1, -1):
            s, c = specs[i], cs[i]
            if s.drop is not None and not masked:
                dh = ops.dropout_apply(dh, *s.drop)          # what came in is the gradient of the masked output
            prev = specs[i - 1] if i > 0 else None
            need_dx = i > 0 or ctx.needs_input_grad[0]
            # a previous block whose mask went into its forward GEMM keeps the masked activation: its act' (a function of
            # the sign) and the mask itself both ride in this block's dx GEMM; the row-dot layer's backward has no mask
            prev_masked = prev is not None and prev.drop is not None and cs[i - 1].drop_fused and not c.one
            fuse_prev = (prev is not None and not prev.norm and prev.act in (1, 2, 3)
                         and (prev.drop is None or prev_masked)
                         and dest[5 * (i - 1) + 1] is not None and c.x.shape[1] == c.k)
            gW, gb = dest[5 * i], dest[5 * i + 1]
            r = mlp_block_bwd(dh, c, need_dx=need_dx, dz_ready=dz_ready, dW_out=gW, db_out=gb,
                              mulsrc=cs[i - 1].h if fuse_prev else None, mul_act=prev.act if fuse_prev else 0,
                              defer_reduce=cap.splitk if (sinks and sunk[5 * i]) else None,
                              colsum_parts=parts if (fuse_prev or c.one) else None,
                              dbp_out=dest[5 * (i - 1) + 1] if fuse_prev else None, db_done=db_done,
                              dx_dropout=prev.drop if (fuse_prev and prev_masked) else None, db_parts=shared)
            masked = fuse_prev and prev_masked
            grads[5 * i + 2], grads[5 * i + 3], grads[5 * i + 4] = r["dgamma"], r["dbeta"], r["dslope"]
            dh = r["dx"]
            # the previous block's bias gradient is on its way if this block's dx launch left its partial sums
            db_done = bool(r.get("dbp_left"))
            dz_ready = fuse_prev
        if shared is None:
            parts.finish()
        if ctx.out_is_h:
            cs[-1].h = None
        for k in range(5 * n):
            if dest[k] is not None and not sunk[k]:
                grads[k] = dest[k]
        return (dh, None, *grads)


def mlp_stack(x2d, blocks):
    """blocks: the MLPBlock modules of the run, in order."""
    global _dropout_calls
    cap = TableCapture.active
    call_dev = getattr(cap, "rng_step", None) if cap is not None else None
    specs, params = [], []
    for b in blocks:
        gamma, beta, rm, rv, eps = b._norm_args()
        slope = b.activation_func.weight if b._act == ACT_CODES["prelu"] else None
        drop = None
        if b.dropout.p and b.training:
            call_no = 0
            if call_dev is None:
                _dropout_calls += 1
                call_no = _dropout_calls
            drop = (float(b.dropout.p), torch.initial_seed(), int(b.dropout.salt), call_no, call_dev)
        specs.append(BlockSpec(b._act, b.normalization or None, b.training, rm, rv, eps, drop))
        params += [b.fc.weight, b.fc.bias, gamma, beta, slope]
    return _MLPStackFn.apply(x2d, specs, *params)


class _DropoutFn(torch.autograd.Function):
    """nn.Dropout in training mode (Layers.py:33-37,46) on the HIP kernel: the mask is a function of
    (seed, salt, call), so the backward pass regenerates it instead of keeping it."""

    @staticmethod
    def forward(ctx, x, p, seed, salt, call_no, call_dev):
        ctx.args = (p, seed, salt, call_no, call_dev)
        return ops.dropout_apply(x.contiguous(), p, seed, salt, call_no, call_dev)

    @staticmethod
    def backward(ctx, g):
        p, seed, salt, call_no, call_dev = ctx.args
        return ops.dropout_apply(g.contiguous(), p, seed, salt, call_no, call_dev), None, None, None, None, None


def dropout(x, p: float, salt: int):
    """Training-mode dropout of x.  Inside a trainer's captured step the trainer's device step counter numbers the
    call (a replayed HIP graph draws a fresh mask every step); elsewhere a host-side call counter does."""
    global _dropout_calls
    cap = TableCapture.active
    call_dev = getattr(cap, "rng_step", None) if cap is not None else None
    call_no = 0
    if call_dev is None:
        _dropout_calls += 1
        call_no = _dropout_calls
    return _DropoutFn.apply(x, float(p), torch.initial_seed(), int(salt), call_no, call_dev)


class _BCEHeadFn(torch.autograd.Function):
    """mean binary cross-entropy of sigmoid(z) against y -- torch.sigmoid + nn.BCELoss (model/LR/LR.py:32,
    model/utils/BaseModel.py:74-77,145) -- with dL/dz from the same launch (torch's -100 clamp of the logarithms)."""

    @staticmethod
    def forward(ctx, z, y, assume_grad):
        total = torch.empty(1, dtype=torch.float32, device=z.device) if assume_grad is not None else None
        r = ops.bce_head([z.contiguous().view(-1)], y, want_dz=True,
                         grad_scale=1.0 if assume_grad is None else float(assume_grad), sum_dz_out=total)
        # dL/dz is final only when the incoming gradient is known in advance; then its batch sum is on offer
        _HeadSum.dz, _HeadSum.total = (r["dz"], total) if assume_grad is not None else (None, None)
        ctx.dz = r["dz"]
        ctx.shape = z.shape
        ctx.assumed = assume_grad is not None
        return r["loss"].view(())

    @staticmethod
    def backward(ctx, g):
        dz = ctx.dz if ctx.assumed else ctx.dz * g
        return dz.view(ctx.shape), None, None


def bce_of_logits(z, y, assume_grad=None):
    """z: (B,) or (B, 1) logits; y: (B,) float32 or int64 labels.
    assume_grad: the caller promises that the gradient arriving at the loss in backward() is this number (1.0 when
    the loss itself is differentiated); dL/dz then leaves the forward launch already scaled and backward() adds no
    launch of its own."""
    return _BCEHeadFn.apply(z, y, assume_grad)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _AddFn.apply(a, b)


class _ResidualFn(torch.autograd.Function):
    """identity (or downsample(x)) + act(norm(fc(x)))   (Layers.py:84-94)."""

    @staticmethod
    def forward(ctx, x2d, W, b, gamma, beta, Wd, bd, act, norm, training, running_mean, running_var, eps):
        h, c = mlp_block_fwd(x2d, W, b, act, norm, gamma, beta, None, training, running_mean, running_var, eps)
        ident = x2d if Wd is None else ops.gemm(x2d, Wd, bias=bd)
        ctx.c = c
        ctx.Wd = Wd
        return ops.add(ident, h)

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        c = ctx.c
        r = mlp_block_bwd(dout, c, need_dx=True)
        dWd = dbd = None
        if ctx.Wd is None:
            dx = ops.add(r["dx"], dout)
        else:
            dWd = ops.gemm(dout, c.x, a_mc=True, b_nc=True)
            dbd = ops.colsum(dout)
            dx = ops.add(r["dx"], ops.gemm(dout, ctx.Wd, b_nc=True))
        return (dx, r["dW"], r["db"], r["dgamma"], r["dbeta"], dWd, dbd, None, None, None, None, None, None)


def residual_block(x2d, W, b, gamma, beta, Wd, bd, act, norm, training, running_mean, running_var, eps=1e-5):
    return _ResidualFn.apply(x2d, W, b, gamma, beta, Wd, bd, act, norm, training, running_mean, running_var, eps)


# -------------------------------------------------------------------------------------------------
# interactions
# -------------------------------------------------------------------------------------------------
class _FMFn(torch.autograd.Function):
    """(B, 1) = 0.5 * sum_k((sum_f e)^2 - sum_f e^2)   (model/FM/FM.py:27-30)."""

    @staticmethod
    def forward(ctx, emb):
        emb = emb.contiguous()
        r = ops.fm_fwd(emb, want_sum=True, want_fm=True)
        ctx.save_for_backward(emb, r["sum"])
        return r["fm"].view(-1, 1)

    @staticmethod
    def backward(ctx, g):
        emb, S = ctx.saved_tensors
        return ops.fm_bwd(emb, S, dfm=g.contiguous().view(-1))


def fm_second_order(emb):
    return _FMFn.apply(emb)


class _BiInteractionFn(torch.autograd.Function):
    """(B, d) = 0.5 * ((sum_f e)^2 - sum_f e^2): textbook NFM Bi-Interaction pooling."""

    @staticmethod
    def forward(ctx, emb):
        emb = emb.contiguous()
        r = ops.fm_fwd(emb, want_sum=True, want_bi=True, want_fm=False)
        ctx.save_for_backward(emb, r["sum"])
        return r["bi"]

    @staticmethod
    def backward(ctx, g):
        emb, S = ctx.saved_tensors
        return ops.fm_bwd(emb, S, dbi=g.contiguous())


def bi_interaction(emb):
    return _BiInteractionFn.apply(emb)
