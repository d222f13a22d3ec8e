"""Fused training step: the whole of BaseModel.train_step (model/utils/BaseModel.py:141-150) as one
sequence of HIP kernels with no host synchronisation, optionally replayed from a HIP graph.

    K1  fused gather (+ LR sum, + FM sum-square)         rs_gather_lr_fm_fwd
    MLP forward GEMMs with bias+activation epilogues     rs_gemm_f32 (...)
    loss head: sigmoid + BCE + dL/dlogit                 rs_bce_head
    MLP backward GEMMs (dW split over the batch)         rs_gemm_f32, rs_colsum_f32
    K3  sort / segment / reduce of embedding gradients   rs_sparse_bwd_sort, rs_sparse_bwd_reduce
    global-norm clip coefficient (device scalar)         rs_sumsq_partials, rs_clip_coef
    optimizer: dense towers + touched embedding rows     rs_opt_prepare, rs_dense_update, rs_sparse_update

Two embedding-update modes:

* ``mode="sparse"`` (default, the fast path): optimizer state and weights of touched rows only.
  Identical to the reference's dense optimizers for SGD and Adagrad.  For Adam/AdamW a row absent
  from a batch still moves in the reference (zero gradient, decaying moments, AdamW decay): with
  ``exact_adam=True`` (default) every table carries a per-row step counter and the skipped
  zero-gradient steps of a row are replayed right before it is next read (rs_adam_catchup), so
  lookups, gradients and updates are those of the reference's dense optimizer; ``flush()`` brings
  every row up to date (evaluate / state_dict / a learning-rate change do it).  ``exact_adam=False``
  is the "lazy" variant (absent rows keep their weights and moments).  ``emb_regular`` (an L2 term on
  the tables gives every row a gradient) is carried for SGD on one GPU -- the rows off the batch only
  shrink, by the same factor, which lives in one scalar (ops.TableShrink) -- and refused otherwise.
* ``mode="dense"``: the reference's exact semantics -- row gradients are expanded into full-table
  gradients, L2 terms are added, and dense optimizer kernels sweep every table.  Used for
  bit-for-semantics parity tests and as the AUC reference for the sparse mode.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional

import torch

from . import functional as Fn
from . import ops
from ._lib import RSError
from .layers import MLPBlock
from .ops import ACT_CODES

_FUSABLE = (0, 1, 2, 3)
N_DENSE_PARTIALS = 256


class _Fork:
    """Run the enclosed launches on a side stream, ordered after the work already queued on the
    current stream, with their scratch taken from the side stream's own workspace."""

    def __init__(self, side, ws):
        self.side, self.ws = side, ws

    def __enter__(self):
        self.side.wait_stream(torch.cuda.current_stream())
        self._sctx = torch.cuda.stream(self.side)
        self._sctx.__enter__()
        self._wctx = ops.use_workspace(self.ws)
        self._wctx.__enter__()

    def __exit__(self, *exc):
        self._wctx.__exit__(*exc)
        self._sctx.__exit__(*exc)
        return False


class FusedTrainer:
    # what a step does for the batch announced with next_idx: "plan" = its index sort and Adam catch-up (this class);
    # subclasses with row-sharded tables exchange its keys instead ("keys")
    _ahead_kind = "plan"

    def __init__(self, model, mode: str = "sparse", max_norm: float = 10.0, use_graph: bool = False,
                 optimizer_type: Optional[str] = None, lr: Optional[float] = None, dist=None, overlap: bool = False,
                 exact_adam: bool = True, betas=(0.9, 0.999)):
        """betas: Adam / AdamW coefficients (torch.optim's defaults, which are what the reference's optimizers run with,
        model/utils/BaseModel.py:64-72)."""
        self.betas = (float(betas[0]), float(betas[1]))
        kind = getattr(model, "KIND", None)
        if kind == "pnn" and not (getattr(model, "pnn_type", None) == "inner"
                                   and ops.pnn_inner_cat_ok(len(model.embedding_layer.lookup_columns),
                                                            model.embedding_layer.embedding_dim)):
            raise RSError("FusedTrainer: of the PNN variants only pnn_type='inner' is scheduled here "
                          "(use_sparse_step runs the others)")
        if kind not in ("lr", "fm", "deepfm", "dnn", "widedeep", "dcn", "pnn"):
            raise RSError(f"FusedTrainer: model kind {kind!r} is not covered by the fused step")
        if mode not in ("sparse", "dense"):
            raise ValueError("mode must be 'sparse' or 'dense'")
        self.model = model
        self.kind = kind
        self.mode = mode
        self.max_norm = float(max_norm)
        self.opt = optimizer_type or model.optimizer_type
        if self.opt not in ops.OPT_CODES:
            raise ValueError("Invalid value for optimizer")
        self.lr = float(lr if lr is not None else model.learning_rate)
        self.emb_reg = model.emb_regular or 0.0
        self.net_reg = model.net_regular or 0.0
        # emb_regular puts the gradient lambda * w on EVERY table row.  Under SGD on one GPU that is a uniform shrink of
        # the rows off the batch, which the touched-rows step carries in one scalar (ops.TableShrink); under Adam /
        # Adagrad every row moves by its own amount in every step and only the dense sweep serves it.
        self.shrink = None
        self._shrink_mode = mode == "sparse" and bool(self.emb_reg)
        if self._shrink_mode and (self.opt != "SGD" or type(self) is not FusedTrainer):
            raise RSError("FusedTrainer(mode='sparse'): emb_regular puts a gradient on every table row; the touched-rows "
                          "step carries it for SGD on one GPU only -- use mode='dense' or turn emb_regular off")
        self.dist = dist
        # dense Adam/AdamW semantics on the touched-rows path: replay a row's skipped zero-gradient steps before it is read
        self.exact = bool(exact_adam) and mode == "sparse" and self.opt in ("Adam", "AdamW")
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RSError("FusedTrainer: the model must live on the MI355X (there is no CPU path)")
        self.device = dev
        self.ws = ops.Workspace(dev)

        self.has_lr = kind in ("lr", "fm", "deepfm", "widedeep")
        self.has_emb = kind != "lr"
        self.has_fm = kind in ("fm", "deepfm")
        self.lrm = getattr(model, "lr_layer", None) or getattr(model, "wide", None) if self.has_lr else None
        stack = getattr(model, "deep", None) or getattr(model, "mlp", None)
        self.blocks: List[MLPBlock] = list(stack) if (stack is not None and kind in ("deepfm", "dnn", "widedeep", "dcn", "pnn")) else []
        # PNN with inner products (model/PNN/PNN.py:58,80-89): the tower reads [embeddings | <e_i, e_j> of all pairs]
        self.pnn_inner = kind == "pnn"
        # DCN (model/DCN/DCN.py:48-52): the cross network beside the tower on the flattened embeddings, a 1-wide Linear
        # over [cross output | tower output]
        self.cross = model.cross if kind == "dcn" else None
        self.fc2 = model.fc if kind == "dcn" else None
        if kind == "dcn" and not self.blocks:
            raise RSError("FusedTrainer: DCN without a deep tower is not scheduled here (use_sparse_step runs it)")
        for b in self.blocks:
            if not isinstance(b, MLPBlock):
                raise RSError("FusedTrainer: the deep tower must be a stack of MLPBlock")

        emb = model.embedding_layer
        self.cols = emb.lookup_columns
        self.F = len(self.cols)
        self.d = emb.embedding_dim if self.has_emb else 1
        self.tab = emb._tab if self.has_emb else None
        self.tab1 = self.lrm.scalar._tab1 if self.has_lr else None
        self.layout = (self.tab or self.tab1).layout
        if self.layout is None:
            raise RSError("FusedTrainer: embedding tables are not on the device")
        # tables the reference owns but never looks up (they matter only under L2 regularisation)
        self.unused = []
        if self.emb_reg:
            if self.has_emb:
                self.unused.append(emb._tab1)
            else:
                self.unused += [emb._tab, emb._tab1]
            if self.has_lr:
                self.unused.append(self.lrm.scalar._tab)

        last = self.blocks[-1] if self.blocks else None
        self._head_fast = bool(last is not None and last.fc.out_features == 1 and not last.normalization
                               and last._act in _FUSABLE and last.fc.in_features % 4 == 0
                               and last.fc.in_features <= 1024 and not last.dropout.p)
        self._flatten_dense()
        self._alloc_state()
        self._plans: Dict[int, ops.SparsePlan] = {}
        self.use_graph = use_graph
        # 1/True: index sort and weight-gradient GEMMs on side streams; 2: only the index sort
        self.overlap = int(overlap)
        self._sort_can_fork = True
        self._update_shares_launch = False     # subclasses whose _embed_update(with_dense=True) does both optimizer steps
        self._reduce_carries_finishes = False  # subclasses whose _embed_backward takes the pending finishes along
        self.side_w = torch.cuda.Stream(device=dev)
        self.side_sort = torch.cuda.Stream(device=dev)
        self.ws_w = ops.Workspace(dev)
        self.ws_sort = ops.Workspace(dev)
        self._graphs = {}
        self._graph_pool = None
        self._capture_error_mode = "global"
        # the base hooks run the whole step on this GPU; subclasses that put collectives between the pieces turn it off
        self._fuse_small = type(self) is FusedTrainer and not self._shrink_mode
        # LDS ring stages of the grouped dW + dX launch of a block's backward (0: one launch per product)
        self.group_gemms = ops.GROUP_STAGES
        self.last = {}
        # device-side error flags (index out of range; on several GPUs also bucket overflow of the exchange) are read
        # every check_every steps -- a host synchronisation -- and by check() before tables are read as a whole
        self.check_every = 128
        self._unchecked = 0
        self._watch = None
        self._first_poll = True          # the first step polls too: the one-time cost of the poll's own kernels falls there
        # Work of the NEXT batch done inside a step (step_tensors(..., next_idx=...)): which tensor it was done for and
        # which of the two plans holds its sort.  None: the next step does that work itself, in front of its lookup.
        self._ahead = None

    # ------------------------------------------------------------------------------------------
    def _dense_params(self):
        ps = []
        if self.has_lr:
            ps.append(self.lrm.bias)
        for b in self.blocks:
            ps += [b.fc.weight, b.fc.bias]
            if b.normalization:
                ps += [b.norm_layer.weight, b.norm_layer.bias]
            if b._act == ACT_CODES["prelu"]:
                ps.append(b.activation_func.weight)
        if self.cross is not None:
            # w_0 .. w_{L-1}, b_0 .. b_{L-1} side by side ((D, 1) each, D a multiple of 4: no alignment gaps): the cross
            # kernels read them as [L, D] blocks of the flat buffer and leave (dW, dB) in the same layout
            ps += list(self.cross.cross_w) + list(self.cross.cross_b)
            ps += [self.fc2.weight, self.fc2.bias]
        return ps

    def _flatten_dense(self):
        """All dense parameters in one flat fp32 buffer (one optimizer launch, one all-reduce);
        the nn.Parameters become views into it."""
        ps = self._dense_params()
        n = sum(p.numel() for p in ps)
        # A weight whose rows are not a multiple of four floats long (PNN-inner's first layer: F*d + F(F-1)/2 inputs) is
        # kept with its rows padded to the next multiple: the padding columns are zero, receive zero gradients and stay
        # zero under every optimizer here, the parameter itself is the strided view without them, and the GEMMs take the
        # padded block (aligned rows) together with an input whose padding columns are zero as well.
        self._wfull, self._gfull = {}, {}
        padded = {}
        if getattr(self, "pnn_inner", False) and self.blocks:
            w0 = self.blocks[0].fc.weight
            if w0.shape[1] % 4:
                padded[id(w0)] = (w0.shape[0], w0.shape[1], (w0.shape[1] + 3) // 4 * 4)
        # keep every view 16-byte aligned so the GEMM loaders can use 16-byte loads
        offs, o = [], 0
        for p in ps:
            offs.append(o)
            o += padded[id(p)][0] * padded[id(p)][2] if id(p) in padded else (p.numel() + 3) // 4 * 4
        self.flat_w = torch.zeros(max(o, 4), dtype=torch.float32, device=self.device)
        # four spare floats behind the gradients: the multi-GPU trainer sends its scalar reductions along
        # with the one all-reduce of this buffer
        self.flat_g_ext = torch.zeros(self.flat_w.numel() + 4, dtype=torch.float32, device=self.device)
        self.flat_g = self.flat_g_ext[:self.flat_w.numel()]
        self.gviews = {}
        with torch.no_grad():
            for p, off in zip(ps, offs):
                if id(p) in padded:
                    rows, k, ld = padded[id(p)]
                    full = self.flat_w[off:off + rows * ld].view(rows, ld)
                    full[:, :k].copy_(p.data)
                    p.data = full[:, :k]
                    gfull = self.flat_g[off:off + rows * ld].view(rows, ld)
                    self._wfull[id(p)], self._gfull[id(p)] = full, gfull
                    self.gviews[id(p)] = gfull[:, :k]
                    continue
                v = self.flat_w[off:off + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                self.gviews[id(p)] = self.flat_g[off:off + p.numel()].view_as(p)
        self.n_dense = o

    def _w(self, b):
        """The weight block the GEMMs of block b read (rows padded to a multiple of four floats where the parameter's
        are not) and where its gradient goes."""
        p = b.fc.weight
        return self._wfull.get(id(p), p.data)

    def _gw(self, b):
        p = b.fc.weight
        return self._gfull.get(id(p), self.gviews[id(p)])

    def _alloc_state(self):
        dev = self.device
        adam = self.opt in ("Adam", "AdamW")
        need_m = adam or self.opt == "Adagrad"
        z = lambda t: torch.zeros_like(t)
        self.flat_m = z(self.flat_w) if need_m else None
        self.flat_v = z(self.flat_w) if adam else None
        self.tables = []           # (stacked weight, m, v, dense grad or None, used?)
        for tab, used in ([(self.tab, True)] if self.tab else []) + ([(self.tab1, True)] if self.tab1 else []) + \
                [(t, False) for t in self.unused]:
            w = tab.stacked
            g = z(w) if self.mode == "dense" else None
            self.tables.append({"tab": tab, "w": w, "m": z(w) if need_m else None, "v": z(w) if adam else None,
                                "g": g, "used": used})
        wd = 0.01 if self.opt == "AdamW" else 0.0
        eps = 1e-10 if self.opt == "Adagrad" else 1e-8
        self.hyper = ops.make_hyper(dev, self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=eps, weight_decay=wd)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.scal = torch.tensor([0.0, 1.0], dtype=torch.float32, device=dev)
        self._versions = [(t["tab"], t["tab"].version) for t in self.tables]
        for t in self.tables:                # an earlier trainer's pending steps, then its counters go
            t["tab"].flush_rows()
            t["tab"].row_state = None
        if self.exact:
            self._attach_row_state()
        if self._shrink_mode:
            # one scale for every table the reference regularises, looked up by the model or not
            self.shrink = ops.TableShrink([t["w"] for t in self.tables], self.emb_reg, dev)
            for t in self.tables:
                t["tab"].row_state = self.shrink

    def _attach_row_state(self):
        """Per-row step counters of the looked-up tables; readers of the tables outside this engine (module forward,
        evaluate, state_dict) find them on the holder and bring rows up to date before reading."""
        bctab = None
        for t in self.tables:
            if not t["used"]:
                continue
            rs = ops.RowAdam(t["w"], t["m"], t["v"], self.opt, self.hyper, self.step_count, bctab=bctab)
            bctab = rs.bctab
            t["rs"] = t["tab"].row_state = rs

    def _row_states(self):
        te, t1 = self._table_entry(self.tab), self._table_entry(self.tab1)
        return (te or {}).get("rs"), (t1 or {}).get("rs")

    def flush(self):
        """Bring every table row up to the current step (only the exact Adam mode leaves rows behind)."""
        for t in self.tables:
            if t.get("rs") is not None:
                t["rs"].flush()
        if self.shrink is not None:
            self.shrink.flush()

    def check(self):
        """Raise IndexError if a kernel saw an out-of-range index since the last check (the reference's nn.Embedding
        raises at once; the kernels clamp such indices to the padding row and set a flag).  Device sync."""
        self._unchecked = 0
        self.layout.check_indices()

    def _flag_tensors(self):
        """(device flag, what it means) pairs that check() would look at."""
        return [(self.layout.err_flag, "index")]

    def _poll_flags(self):
        """The periodic form of check(): no host synchronisation (ops.FlagWatch) -- a raised flag is reported by the
        synchronous check() one to two polling intervals after the step that raised it."""
        self._unchecked = 0
        if self._watch is None:
            self._watch = ops.FlagWatch(self.device)
        if self._watch.poll(self._flag_tensors()):
            self.check()

    def _check_tables(self):
        for tab, ver in self._versions:
            if tab.version != ver:
                raise RSError("FusedTrainer: the model's tables were moved to other storage after the trainer was "
                              "built (model.to(another device)?); build a new trainer")

    def _table_entry(self, tab):
        for t in self.tables:
            if t["tab"] is tab:
                return t
        return None

    def decay_lr(self, factor=0.1, min_lr=1e-6):
        self.flush()                 # rows left behind are replayed with the rate their skipped steps had
        self.lr = max(self.lr * factor, min_lr)
        self.hyper[0:1].fill_(self.lr)
        return self.lr

    def plan(self, B: int, slot: int = 0) -> ops.SparsePlan:
        """Sort plan + reduction buffers for batches of B samples.  Two slots: while a step works with one, the sort of
        the next batch may already be written into the other (step_tensors(..., next_idx=...))."""
        key = (B, slot)
        if key not in self._plans:
            p = ops.SparsePlan(self.layout, cap=B, d=self.d, with_lr=self.has_lr)
            n_extra = N_DENSE_PARTIALS * (1 + len(self.tables))
            p.partials = torch.zeros(p.n_partials + n_extra, dtype=torch.float32, device=self.device)
            self._plans[key] = p
        return self._plans[key]

    # ---- work of the next batch done inside a step -------------------------------------------------------------
    def _ahead_supported(self, B: int) -> bool:
        """The sort of the next batch can ride in a backward GEMM launch and its Adam catch-up beside the update."""
        return (self._fuse_small and not self.overlap and not self.net_reg and self.mode == "sparse"
                and B <= ops.group_sort_max_entries())

    @staticmethod
    def _tensor_key(t: torch.Tensor):
        return (t.data_ptr(), t._version, tuple(t.shape))

    def _ahead_slot_for(self, idx: torch.Tensor):
        """Slot of the plan that already holds the sort (and catch-up) of `idx`, or None."""
        a = self._ahead
        if a is not None and a["key"] == self._tensor_key(idx):
            return a["slot"]
        return None

    def _cancel_ahead(self):
        """The batch announced with next_idx did not come: take back the promise its rows carry."""
        a, self._ahead = self._ahead, None
        if a is not None and a["marked"]:
            with ops.use_workspace(self.ws):
                self.plan(a["B"], a["slot"]).catchup_plan(*self._row_states(), mode=1)

    # ------------------------------------------------------------------------------------------
    def forward_parts(self, idx: torch.Tensor, defer_head: bool = False, sort_plan=None, mark_ahead: int = 0):
        """Forward pass; returns (logit parts, saved context).  defer_head: leave a 1-wide last block to the
        caller (the training step fuses it with the loss head); then ctx["head_in"] is its input.  sort_plan: also
        sort the batch into this plan, in the gather's launch.  mark_ahead=1: this is the forward of a training step
        (the rows read here are stepped at its end)."""
        B = idx.shape[1]
        ctx = {"idx": idx, "B": B}
        parts = []
        if sort_plan == "done":
            r = self._embed_forward(idx)          # sorted and caught up during the step before: the lookup alone
        elif self.exact and sort_plan is not None and mark_ahead == 0:
            # the replay of skipped Adam steps must precede the lookup: it shares the sort's launch, the gather follows
            # (the sort is cut in two where it can be: its first passes here, the rest beside the gather)
            sp = sort_plan.split_passes_for(B)
            sort_plan.sort_and_catchup(idx, *self._row_states(), split_passes=sp)
            r = self._embed_forward(idx) if sp == 0 else self._embed_forward_sorting(idx, sort_plan, resume_passes=sp)
        else:
            self._embed_catch_up(idx, mark_ahead)
            r = self._embed_forward(idx) if sort_plan is None else self._embed_forward_sorting(idx, sort_plan)
        ctx["emb"] = r.get("emb")
        ctx["S"] = r.get("sum")
        if self.has_lr:
            parts.append(r["lr"])
        if self.has_fm:
            parts.append(r["fm"])
        if self.blocks:
            # PNN-inner: one launch writes [embeddings | inner products | zeros up to a multiple of four columns]
            h = ops.pnn_inner_cat_fwd(r["emb"]) if self.pnn_inner else r["emb"].view(B, self.F * self.d)
            cs = []
            for bi, b in enumerate(self.blocks):
                if bi == len(self.blocks) - 1 and self._head_fast:
                    if defer_head:
                        cs.append(("head", h, None))
                        ctx["head_in"] = h
                        h = None
                        break
                    # 1-wide head: one wave per row instead of a 64-wide GEMM tile
                    hh = ops.rowdot_fwd(h, b.fc.weight.data.view(-1), b.fc.bias.data, b._act)
                    cs.append(("head", h, hh))
                    h = hh
                    break
                gamma = beta = rm = rv = slope = None
                eps = 1e-5
                if b.normalization:
                    gamma, beta, eps = b.norm_layer.weight.data, b.norm_layer.bias.data, b.norm_layer.eps
                    if b.normalization == "bn":
                        rm, rv = b.norm_layer.running_mean, b.norm_layer.running_var
                if b._act == ACT_CODES["prelu"]:
                    slope = b.activation_func.weight.data
                h, c = Fn.mlp_block_fwd(h, self._w(b), b.fc.bias.data, b._act, b.normalization or None, gamma,
                                        beta, slope, self.model.training, rm, rv, eps,
                                        dropout=self._drop_key(b) if self._drops(b) else None)
                cs.append(c)
                if self._drops(b) and not c.drop_fused:
                    h = self._dropout(h, b)              # c.h stays the activation before the mask (act' needs it)
            ctx["mlp"] = cs
            if self.cross is not None:
                x0 = r["emb"].view(B, self.F * self.d)
                w_all, b_all = self._cross_params()
                xo, sx = ops.dcn_cross_fwd(x0, w_all, b_all)
                logit = ops.rowdot2_fwd(xo, h, self.fc2.weight.data.view(-1), self.fc2.bias.data, 0)
                ctx["dcn"] = (x0, xo, sx, h)
                parts.append(logit)
            elif h is not None:
                parts.append(h.view(B))
        return parts, ctx

    def _cross_params(self):
        """(w [L, D], b [L, D]) of the cross network as blocks of the flat weight buffer, and where their gradients go."""
        L = self.cross.cross_layers_num
        D = self.cross.input_dim
        w0 = self.cross.cross_w[0].data
        off = (w0.data_ptr() - self.flat_w.data_ptr()) // 4
        blk = self.flat_w[off:off + 2 * L * D]
        return blk[:L * D].view(L, D), blk[L * D:].view(L, D)

    def _cross_grad_block(self):
        L, D = self.cross.cross_layers_num, self.cross.input_dim
        off = (self.cross.cross_w[0].data.data_ptr() - self.flat_w.data_ptr()) // 4
        return self.flat_g[off:off + 2 * L * D]

    # ---- embedding hooks (overridden by the row-sharded multi-GPU trainer) -----------------------
    def _embed_catch_up(self, idx, mark_ahead: int):
        """Exact Adam on touched rows: the rows this batch reads are brought up to the current step first."""
        if self.exact:
            main, one = self._row_states()
            ops.adam_catchup(idx, self.layout, main, one, mark_ahead)

    def _embed_forward(self, idx):
        """dict(emb [B,F,d], sum [B,d], lr [B], fm [B]) as needed by the model kind: K1, one launch."""
        table = self.tab.stacked if self.has_emb else None
        table1 = self.tab1.stacked if self.has_lr else None
        bias = self.lrm.bias.data if self.has_lr else None
        return ops.gather_lr_fm_fwd(idx, self.layout, table, table1, bias, want_emb=self.has_emb,
                                    want_sum=self.has_fm, want_lr=self.has_lr, want_fm=self.has_fm,
                                    scale=self.shrink.gscale if self.shrink is not None else None)

    def _embed_forward_sorting(self, idx, plan, resume_passes: int = 0):
        """_embed_forward plus the index sort of the sparse backward (or its remaining passes) in the same launch."""
        table = self.tab.stacked if self.has_emb else None
        table1 = self.tab1.stacked if self.has_lr else None
        bias = self.lrm.bias.data if self.has_lr else None
        return plan.sort_and_gather(idx, table, table1, bias, want_emb=self.has_emb, want_sum=self.has_fm,
                                    want_lr=self.has_lr, want_fm=self.has_fm, resume_passes=resume_passes)

    def _grad_scale(self) -> float:
        return 1.0

    def _embed_backward(self, plan, idx, gout, dz, ctx, sorted_already, dense_partials=None, colsum=None, splitk=None):
        """Row gradients of the touched rows; fills plan.partials[:plan.n_partials] with ||g||^2 partials.
        dense_partials: also leave the ||g||^2 partials of the flat dense gradient there (same launch)."""
        B = idx.shape[1]
        if not sorted_already:
            plan.sort_columnar(idx)
        plan.reduce(plan.partials[:plan.n_partials], gout=gout, dfm=dz if self.has_fm else None, S=ctx["S"],
                    emb=ctx["emb"].view(B * self.F, self.d) if self.has_fm else None,
                    dlr=dz if self.has_lr else None, Fdiv=self.F,
                    dense=self.flat_g if dense_partials is not None else None, dense_partials=dense_partials,
                    colsum=colsum, splitk=splitk)

    def _reduce_dense_grads(self):
        """Multi-GPU: sum the flat dense gradient over ranks.  Single GPU: nothing."""

    def _grad_norm_partials(self, plan, fuse, dense_p):
        """Touched-rows mode, between the embedding backward and the clip coefficient: whatever the dense gradient still
        needs (its sum over ranks, its L2 term, its sum-of-squares partials unless they came out of the embedding
        reduction's launch); returns the buffer of partial sums whose total is the squared global gradient norm."""
        self._reduce_dense_grads()
        if self.net_reg:
            ops.axpy(self.net_reg, self.flat_w, self.flat_g)
        if not fuse:
            ops.sumsq_partials(self.flat_g, dense_p)
        return plan.partials[:plan.n_partials + N_DENSE_PARTIALS]

    def _embed_update(self, plan, with_dense=False, stamps=True, next_plan=None):
        te, t1 = self._table_entry(self.tab), self._table_entry(self.tab1)
        kw = dict(table=te["w"] if te else None, m=te["m"] if te else None, v=te["v"] if te else None,
                  table1=t1["w"].view(-1) if t1 else None, m1=t1["m"].view(-1) if t1 and t1["m"] is not None else None,
                  v1=t1["v"].view(-1) if t1 and t1["v"] is not None else None,
                  rs=(te or {}).get("rs"), rs1=(t1 or {}).get("rs"))
        if next_plan is not None and self.exact:
            plan.update_with_dense_ahead(self.opt, self.hyper, self.scal, self.flat_w, self.flat_g, self.flat_m, self.flat_v,
                                         next_plan, **kw)
            return
        if not stamps:
            kw["rs"] = kw["rs1"] = None
        if with_dense:
            plan.update_with_dense(self.opt, self.hyper, self.scal, self.flat_w, self.flat_g, self.flat_m, self.flat_v, **kw)
        else:
            # (tables kept as w / G take the step with the coefficient c / G)
            plan.update(self.opt, self.hyper, self.shrink.scal_tab if self.shrink is not None else self.scal, **kw)

    def _drops(self, b) -> bool:
        """Block b's output goes through dropout in this pass (Layers.py:33-37,46)."""
        return b is not None and bool(b.dropout.p) and self.model.training

    def _drop_key(self, b):
        """(p, seed, salt, call_no, call_dev) of block b's mask at the current step: the device step counter numbers it,
        so both passes of a step -- and every replay of a captured step -- agree."""
        return (float(b.dropout.p), torch.initial_seed(), int(b.dropout.salt), 0, self.step_count)

    def _dropout(self, x, b):
        """The mask of block b at the current step, applied to an activation (forward) or to a gradient (backward)."""
        return ops.dropout_apply(x, *self._drop_key(b))

    def _bump_bn_counters(self):
        for b in self.blocks:
            if b.normalization == "bn" and self.model.training:
                b.norm_layer.num_batches_tracked.add_(1)

    def logits(self, idx: torch.Tensor) -> torch.Tensor:
        """Forward only.  In training mode BatchNorm running statistics move, as in a module forward."""
        self._bump_bn_counters()
        with ops.use_workspace(self.ws):
            parts, _ = self.forward_parts(idx)
            out = parts[0]
            for p in parts[1:]:
                out = ops.add(out, p)
        return out.view(-1, 1)

    def _fork(self, side, ws):
        """Context: run the enclosed launches on `side`, ordered after everything issued so far on the
        current stream.  Returns immediately; join with current_stream().wait_stream(side)."""
        return _Fork(side, ws)

    def _step_impl(self, idx: torch.Tensor, labels: torch.Tensor, next_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        B = idx.shape[1]
        # Was this batch announced to the step before (next_idx)?  Then its sort and its Adam catch-up are done.
        slot = self._ahead_slot_for(idx)
        done_ahead = slot is not None
        if not done_ahead:
            self._cancel_ahead()
            slot = 0
        self._ahead = None
        do_ahead = (next_idx is not None and self._ahead_supported(B) and tuple(next_idx.shape) == tuple(idx.shape)
                    and next_idx.dtype == torch.int64 and next_idx.is_contiguous())
        plan_mode = self._ahead_kind == "plan"
        plan = self.plan(B, slot if plan_mode else 0)
        next_plan = self.plan(B, 1 - slot) if (do_ahead and plan_mode) else None
        # what the embedding hooks of subclasses need to know about this step
        self._step_ctx = {"slot": slot, "done_ahead": done_ahead, "next_idx": next_idx if do_ahead else None}
        # exact Adam with work done ahead: a batch's rows are stamped "stepped" before its update (catchup_plan), and
        # the update writes no stamps
        promised = plan_mode and self.exact and (done_ahead or do_ahead)
        main = torch.cuda.current_stream()
        overlap = self.overlap == 1
        keep = []                    # intermediates read on side streams stay alive until the join
        # The index sort needs only the batch: it runs beside the forward pass.
        fork_sort = bool(self.overlap) and self._sort_can_fork and not (done_ahead and plan_mode)
        if fork_sort:
            with self._fork(self.side_sort, self.ws_sort):
                plan.sort_columnar(idx)
        head_fast = bool(self.blocks) and self._head_fast
        # single GPU, one stream: the sort shares the gather's launch (it keeps only F workgroups busy)
        sort_early = self._fuse_small and not fork_sort
        if done_ahead and plan_mode:
            parts, ctx = self.forward_parts(idx, defer_head=head_fast, sort_plan="done")
        elif promised:
            # first step of a run that announces its batches: sort, then replay + promise per distinct row, then look up
            plan.sort_columnar(idx)
            plan.catchup_plan(*self._row_states(), mode=0)
            parts, ctx = self.forward_parts(idx, defer_head=head_fast, sort_plan="done")
        else:
            # the update at the end of the step stamps the rows it steps, so the catch-up does not mark ahead
            parts, ctx = self.forward_parts(idx, defer_head=head_fast, sort_plan=plan if sort_early else None)
        sorted_already = fork_sort or sort_early or (plan_mode and done_ahead) or promised
        ahead_sort_pending = do_ahead and plan_mode      # the next batch's sort still has to be issued
        ahead_sort_half = False              # ... its first half has been
        lastb = self.blocks[-1] if self.blocks else None
        # partial column sums -> one finishing launch (bias gradients; with a fused head also the loss and its batch sums)
        csp = ops.ColsumParts(ops.default_workspace(self.device)) if self.blocks else None
        head_prev = self.blocks[-2] if head_fast and len(self.blocks) > 1 else None
        head_fuse_prev = (head_prev is not None and not head_prev.normalization and head_prev._act in (1, 2, 3)
                          and not self._drops(head_prev))
        head_both = head_fast and ctx["head_in"].shape[1] <= 1024
        if head_both:
            # the 1-wide last block, the loss, dL/dlogit and the block's backward in one launch
            head = ops.head_fwd_bwd(ctx["head_in"], lastb.fc.weight.data.view(-1), lastb.fc.bias.data, lastb._act, parts,
                                    labels, csp, dW_out=self.gviews[id(lastb.fc.weight)].view(-1),
                                    grad_scale=self._grad_scale(),
                                    sum_dz_out=self.gviews[id(self.lrm.bias)] if self.has_lr else None,
                                    sum_dzh_out=self.gviews[id(lastb.fc.bias)],
                                    mul_act=head_prev._act if head_fuse_prev else 0,
                                    dbp_out=self.gviews[id(head_prev.fc.bias)] if head_fuse_prev else None)
            parts = parts + [head["h"]]
        elif head_fast:
            # the 1-wide last block, the loss and dL/dlogit in one launch
            head = ops.head_fwd_bce(ctx["head_in"], lastb.fc.weight.data.view(-1), lastb.fc.bias.data, lastb._act, parts,
                                    labels, grad_scale=self._grad_scale(),
                                    sum_dz_out=self.gviews[id(self.lrm.bias)] if self.has_lr else None,
                                    sum_dzh_out=self.gviews[id(lastb.fc.bias)], defer=csp)
            parts = parts + [head["h"]]
        else:
            # (the batch sum of dL/dlogit is the bias gradient of whatever adds a scalar bias to the logit: the LR layer,
            #  or DCN's last Linear)
            head = ops.bce_head(parts, labels, want_dz=True, grad_scale=self._grad_scale(),
                                sum_dz_out=self.gviews[id(self.lrm.bias)] if self.has_lr
                                else self.gviews[id(self.fc2.bias)] if self.fc2 is not None else None)
        loss, dz = head["loss"], head["dz"]
        keep += [parts, ctx, head]

        # ---- dense backward ------------------------------------------------------------------
        # Critical chain on the main stream: dZ_l -> dX -> dZ_(l-1) ...; the weight/bias gradients of
        # a layer depend only on its dZ and run beside the chain on the side stream.
        gout = None
        pending_colsum = pending_splitk = None
        if self.blocks:
            cs = ctx["mlp"]
            dh = dz.view(B, 1)
            dz_ready = False
            top = len(self.blocks) - 1

            def wgrad(fn):
                if overlap:
                    with self._fork(self.side_w, self.ws_w):
                        fn()
                else:
                    fn()

            # weight gradients are read by nobody before the gradient norm: their split-K slabs are reduced together
            skp = ops.SplitKParts(ops.default_workspace(self.device)) if not overlap else None
            # a block's dW and dX need only its dZ and nothing reads either before the next launch: one launch for both
            grp = ops.GemmGroup(self.group_gemms) if (skp is not None and self.group_gemms) else None
            db_done = False              # the bias gradient of the block about to be visited is already on its way
            dxo = None
            if self.cross is not None:
                # backward of Linear(D + h, 1) over [cross output | tower output]: the cross branch's gradient, the tower's
                # (with act' of its last block, whose bias gradient then comes out of the same launch) and dW
                x0, xo, sx, hdeep = ctx["dcn"]
                lastd = self.blocks[top]
                fuse_last = (not lastd.normalization and lastd._act in (1, 2, 3) and not self._drops(lastd))
                dxo, dh = ops.head2_bwd(dz, self.fc2.weight.data.view(-1), xo, hdeep, lastd._act if fuse_last else 0, csp,
                                        dW_out=self.gviews[id(self.fc2.weight)].view(-1),
                                        dbp_out=self.gviews[id(lastd.fc.bias)] if fuse_last else None)
                dz_ready = db_done = fuse_last
                keep.append(dxo)
            if head_fast:
                _, xin, _ = cs[top]
                prev = self.blocks[top - 1] if top > 0 else None
                fuse_prev = (prev is not None and not prev.normalization and prev._act in (1, 2, 3)
                             and not self._drops(prev))
                # dX, dW of the head and (when the previous act' is fused) the previous block's bias gradient
                if head_both:
                    dh = head["dx"]
                else:
                    dh = ops.head_bwd(head["dzh"], lastb.fc.weight.data.view(-1), xin, prev._act if fuse_prev else 0, csp,
                                      dW_out=self.gviews[id(lastb.fc.weight)].view(-1),
                                      dbp_out=self.gviews[id(prev.fc.bias)] if fuse_prev else None)
                if self._drops(prev):
                    dh = self._dropout(dh, prev)         # the head read the masked activation
                    keep.append(dh)
                dz_ready = db_done = fuse_prev
                top -= 1
            if not head_fast and self._drops(self.blocks[top]):
                dh = self._dropout(dh, self.blocks[top])     # the loss read the masked output of the last block
                keep.append(dh)
            for i in range(top, -1, -1):
                b, c = self.blocks[i], cs[i]
                prev = self.blocks[i - 1] if i > 0 else None
                # a previous block whose mask went into its forward GEMM kept the masked activation: the mask and its
                # act' (a function of the sign) both ride in this block's dX GEMM
                prev_masked = self._drops(prev) and cs[i - 1].drop_fused
                fuse_prev = (prev is not None and not prev.normalization and prev._act in (1, 2, 3)
                             and (not self._drops(prev) or prev_masked))
                mask_key = self._drop_key(prev) if (fuse_prev and prev_masked) else None
                if not b.normalization and b._act in _FUSABLE:
                    dzl = dh if (dz_ready or b._act == 0) else ops.act_bwd(dh, b._act, h=c.h)[0]
                    keep.append(dzl)
                    gW, gb = self._gw(b), self.gviews[id(b.fc.bias)]
                    if db_done and grp is not None:
                        grp.add(dzl, c.x, a_mc=True, b_nc=True, out=gW, defer_reduce=skp)
                    elif db_done:
                        wgrad(lambda dzl=dzl, c=c, gW=gW: ops.gemm(dzl, c.x, a_mc=True, b_nc=True, out=gW, defer_reduce=skp))
                    else:
                        wgrad(lambda dzl=dzl, c=c, gW=gW, gb=gb: (ops.gemm(dzl, c.x, a_mc=True, b_nc=True, out=gW,
                                                                           defer_reduce=skp),
                                                                  ops.colsum(dzl, out=gb)))
                    # dX of this block = dZ of the previous one when its act' rides in the epilogue; its column
                    # sums (the previous block's bias gradient) then come out of the same launch
                    if fuse_prev:
                        csp.expect(self.gviews[id(prev.fc.bias)])
                    if grp is not None:
                        dh = grp.add(dzl, c.W, b_nc=True, mulsrc=cs[i - 1].h if fuse_prev else None,
                                     mul_act=prev._act if fuse_prev else 0, colsum_part=csp if fuse_prev else None,
                                     dropout=mask_key)
                    else:
                        dh = ops.gemm(dzl, c.W, b_nc=True, mulsrc=cs[i - 1].h if fuse_prev else None,
                                      mul_act=prev._act if fuse_prev else 0, colsum_part=csp if fuse_prev else None,
                                      dropout=mask_key)
                    if grp is not None:
                        if ahead_sort_pending and self.group_gemms == 3 and i <= 1:
                            # The sort of the next batch rides beside the two longest backward launches, half of it in
                            # each (beside GEMM workgroups a sort workgroup takes several times as long as alone).
                            phase = (2 if ahead_sort_half else 0) if i == 0 else 1
                            grp.launch(sort=(next_plan, next_idx, phase))
                            ahead_sort_half = phase == 1
                            ahead_sort_pending = phase == 1
                        else:
                            grp.launch()
                    db_done = fuse_prev
                else:
                    r = Fn.mlp_block_bwd(dh, c, need_dx=True, dz_ready=dz_ready,
                                         dW_out=self._gw(b), db_out=self.gviews[id(b.fc.bias)],
                                         mulsrc=cs[i - 1].h if fuse_prev else None, mul_act=prev._act if fuse_prev else 0,
                                         dx_dropout=mask_key)
                    if b.normalization:
                        self.gviews[id(b.norm_layer.weight)].copy_(r["dgamma"])
                        self.gviews[id(b.norm_layer.bias)].copy_(r["dbeta"])
                    if r["dslope"] is not None:
                        self.gviews[id(b.activation_func.weight)].copy_(r["dslope"])
                    dh = r["dx"]
                    keep.append(r)
                    db_done = False
                if self._drops(prev) and mask_key is None:
                    dh = self._dropout(dh, prev)         # this block read the masked activation of the previous one
                dz_ready = fuse_prev
                keep.append(dh)
            if self.cross is not None:
                # the cross network's backward: its share of the embeddings' gradient and (dW, dB) as partial rows whose
                # sums join the step's one finishing launch
                x0, xo, sx, hdeep = ctx["dcn"]
                w_all, b_all = self._cross_params()
                dtower = dh
                dh, _, _ = ops.dcn_cross_bwd(x0, sx, dxo, w_all, b_all, g_out=self._cross_grad_block(), parts=csp,
                                             dx_add=dtower.view(B, self.F * self.d))
                keep += [dtower, dh]
            if self.pnn_inner:
                # gradient of [embeddings | inner products | pad] -> gradient of the embeddings
                dcat = dh
                dh = ops.pnn_inner_cat_bwd(ctx["emb"], dcat)
                keep += [dcat, dh]
            # the bias-gradient finishes and the weight-gradient split-K reduces: beside the first launch of the
            # embedding reduction when that is fused
            carry = self._reduce_carries_finishes and not overlap and self.mode == "sparse"
            if carry or (self._fuse_small and not overlap and not self.net_reg and self.mode == "sparse"):
                pending_colsum = csp.hand_over()
                pending_splitk = skp.hand_over() if skp is not None else None
            else:
                if skp is not None:
                    skp.finish()
                csp.finish()
            gout = dh.view(B * self.F, self.d)

        if fork_sort:
            main.wait_stream(self.side_sort)
        if ahead_sort_pending:
            # no GEMM launch to ride in (or none for the second half): a launch of its own
            if ahead_sort_half:
                next_plan._sort_in_group(None, 0, next_idx, 2)
            else:
                next_plan.sort_columnar(next_idx)
        # ---- embedding backward: segment reduce of the sorted batch -----------------------------
        P = plan.partials
        n_sp = plan.n_partials
        dense_p = P[n_sp:n_sp + N_DENSE_PARTIALS]
        # single GPU, no L2 term, everything on one stream: the dense gradient is final here, so its norm partials
        # come out of the embedding reduction's second launch and the two optimizer steps share one launch
        fuse = self._fuse_small and not overlap and not self.net_reg and self.mode == "sparse"
        if fuse:
            self._embed_backward(plan, idx, gout, dz, ctx, sorted_already=sorted_already, dense_partials=dense_p,
                                 colsum=pending_colsum, splitk=pending_splitk)
        elif pending_colsum is not None or pending_splitk is not None:
            self._embed_backward(plan, idx, gout, dz, ctx, sorted_already=sorted_already, colsum=pending_colsum,
                                 splitk=pending_splitk)
        else:
            self._embed_backward(plan, idx, gout, dz, ctx, sorted_already=sorted_already)

        if overlap:
            main.wait_stream(self.side_w)       # every dense gradient is in flat_g from here on
        # ---- L2 terms, global norm, clip coefficient ------------------------------------------
        if self.mode == "sparse":
            norm_in = self._grad_norm_partials(plan, fuse, dense_p)
        else:
            self._reduce_dense_grads()
            if self.net_reg:
                ops.axpy(self.net_reg, self.flat_w, self.flat_g)
            ops.sumsq_partials(self.flat_g, dense_p)
            off = n_sp + N_DENSE_PARTIALS
            for t in self.tables:
                t["g"].zero_()
            te, t1 = self._table_entry(self.tab), self._table_entry(self.tab1)
            plan.scatter_dense(grad=te["g"] if te else None, grad1=t1["g"].view(-1) if t1 else None)
            for t in self.tables:
                if self.emb_reg:
                    ops.axpy(self.emb_reg, t["w"], t["g"])
                ops.sumsq_partials(t["g"], P[off:off + N_DENSE_PARTIALS])
                off += N_DENSE_PARTIALS
            norm_in = P[n_sp:off]
        if self.shrink is not None:
            # the tables' L2 term: 2 lambda <g, w> over the touched rows and lambda^2 ||W||^2 join the norm, the rows off
            # the batch shrink through the scale, lambda/2 ||W||^2 joins the loss
            te, t1 = self._table_entry(self.tab), self._table_entry(self.tab1)
            dot = plan.touched_dot(table=te["w"] if te else None, table1=t1["w"].view(-1) if t1 else None)
            self.shrink.clip_and_prepare(norm_in, n_sp, dot, self.max_norm, self.scal, self.step_count, self.hyper,
                                         loss=loss)
        else:
            ops.clip_and_prepare(norm_in, self.max_norm, self.scal, self.step_count, self.hyper)
        if self.net_reg or (self.emb_reg and self.mode == "dense"):
            loss = self._add_reg_to_loss(loss)

        # ---- optimizer --------------------------------------------------------------------------
        both = fuse or (self._update_shares_launch and self.mode == "sparse")     # dense and touched-row steps in one launch
        if promised:
            # the rows of this batch carry their stamp already; beside the update, the rows of the next batch are
            # brought up to date and promised to the next update
            self._embed_update(plan, with_dense=True, stamps=False, next_plan=next_plan)
        elif both:
            self._embed_update(plan, with_dense=True)
        else:
            ops.dense_update(self.flat_w, self.flat_g, self.flat_m, self.flat_v, self.opt, self.hyper, self.scal)
        if both or promised:
            pass
        elif self.mode == "sparse":
            self._embed_update(plan)
        else:
            for t in self.tables:
                ops.dense_update(t["w"].view(-1), t["g"].view(-1), t["m"].view(-1) if t["m"] is not None else None,
                                 t["v"].view(-1) if t["v"] is not None else None, self.opt, self.hyper, self.scal)
        if do_ahead:
            self._ahead = {"key": self._tensor_key(next_idx), "slot": 1 - slot, "B": B, "marked": self.exact and plan_mode}
        self.last = {"dz": dz, "parts": parts, "plan": plan}
        del keep
        return loss

    def gradients(self):
        """name -> gradient of the last step as the reference's train_step leaves it in p.grad (model/utils/
        BaseModel.py:141-150: backward of loss + regularization, then clip_grad_norm_): every dense parameter and every
        table, dense (touched-rows mode: the tables the step looked up, zeros but for the rows of the last batch --
        that mode takes no embedding L2 term).  A checking aid: it allocates and copies."""
        plan = self.last.get("plan")
        if plan is None:
            raise RSError("FusedTrainer.gradients: no step has run")
        if self.shrink is not None:
            raise RSError("FusedTrainer.gradients: with emb_regular on the touched-rows step the tables' gradients "
                          "(lambda * w of the weights before the step) are never formed; use mode='dense' to look at them")
        coef = self.scal[1]
        out = {}
        dense = {id(p): n for n, p in self.model.named_parameters()}
        for pid, g in self.gviews.items():
            if pid in dense:
                out[dense[pid]] = g * coef
        te, t1 = self._table_entry(self.tab), self._table_entry(self.tab1)
        if self.mode == "dense":
            tabs = list(self.tables)             # tables the step does not look up carry their L2 term
            full = {id(t): t["g"] for t in tabs}
        else:
            tabs = [t for t in (te, t1) if t]
            full = {id(t): torch.zeros_like(t["w"]) for t in tabs}
            plan.scatter_dense(grad=full[id(te)] if te else None, grad1=full[id(t1)].view(-1) if t1 else None)
        for t in tabs:
            st, G = t["w"], full[id(t)]
            row_bytes = st.element_size() * st.shape[1]
            for n, p in self.model.named_parameters():
                off = p.data_ptr() - st.data_ptr()
                if p.dim() == 2 and p.shape[1] == st.shape[1] and 0 <= off < st.numel() * st.element_size() \
                        and off % row_bytes == 0 and p.device == st.device:
                    r = off // row_bytes
                    out[n] = G[r:r + p.shape[0]] * coef
        return out

    def _add_reg_to_loss(self, loss):
        """loss += (net/2)*||dense||^2 + (emb/2)*sum ||table||^2, evaluated on the pre-step weights."""
        segs, coefs = [], []
        bufs = []
        if self.net_reg:
            bufs.append((self.flat_w, self.net_reg / 2))
        if self.emb_reg and self.mode == "dense":
            for t in self.tables:
                bufs.append((t["w"].view(-1), self.emb_reg / 2))
        P = self.ws.get("reg_partials", N_DENSE_PARTIALS * len(bufs))
        for i, (buf, c) in enumerate(bufs):
            ops.sumsq_partials(buf, P[i * N_DENSE_PARTIALS:(i + 1) * N_DENSE_PARTIALS])
            segs.append((i + 1) * N_DENSE_PARTIALS)
            coefs.append(c)
        key = ("reg_desc", tuple(segs), tuple(coefs))
        if getattr(self, "_reg_key", None) != key:
            self._reg_seg = torch.tensor(segs, dtype=torch.int32, device=self.device)
            self._reg_coef = torch.tensor(coefs, dtype=torch.float32, device=self.device)
            self._reg_key = key
        ops.call("rs_loss_add_reg", ops.ptr(loss), ops.ptr(P), ops.ptr(self._reg_seg), ops.ptr(self._reg_coef),
                 len(bufs), ops.stream())
        return loss

    # ------------------------------------------------------------------------------------------
    def step_tensors(self, idx: torch.Tensor, labels: torch.Tensor, resident: bool = False,
                     next_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One training step on device-resident inputs: idx [F, B] int64, labels [B] int64/float32.

        With use_graph, the step is replayed from a HIP graph.  resident=False copies the batch into
        the graph's static input buffers first (what a host feeder does with its one H2D copy per
        step).  resident=True promises that these exact tensors stay alive and are re-used (a ring of
        pre-staged batches): one graph is captured per distinct (idx, labels) buffer pair and
        replayed without any copy.

        next_idx announces the index matrix of the NEXT call (the same tensor object, contents unchanged until then;
        a feeder that stages batches ahead knows it).  Everything that depends on nothing but the batch -- the index
        sort of the sparse backward and, with exact Adam, the replay of the skipped steps of its rows -- is then done
        inside this step, beside its GEMMs and its update, and the next step starts with the lookup.  A promise that
        is not kept costs one small launch (the rows' stamps are taken back); results do not depend on next_idx."""
        self._check_tables()
        if self.shrink is not None:
            self.shrink.stepped()
        self._unchecked += 1
        if self.check_every and (self._unchecked >= self.check_every or self._first_poll):
            self._first_poll = False
            self._poll_flags()
        if self._ahead is not None and self._ahead_slot_for(idx) is None:
            self._cancel_ahead()
        if not self.use_graph:
            with ops.use_workspace(self.ws):
                return self._step_impl(idx, labels, next_idx)
        if resident:
            key = self._graph_key(idx, labels, next_idx, self._ahead)
            ent = self._graphs.get(key)
            if ent is None:
                ent = self._capture(idx, labels, next_idx, self._ahead)
                self._graphs[key] = ent
            ent["graph"].replay()
            self._ahead = dict(ent["ahead_after"]) if ent["ahead_after"] else None
            return ent["loss"]
        self._cancel_ahead()
        key = ("static", tuple(idx.shape), labels.dtype)
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._capture(idx.clone(), labels.clone())
            self._graphs[key] = ent
        ent["idx"].copy_(idx)
        ent["labels"].copy_(labels)
        ent["graph"].replay()
        return ent["loss"]

    def _graph_key(self, idx, labels, next_idx, entry):
        """A captured step depends on its buffers, on the buffer it prepares the next step for, and on whether (and in
        which plan) its own batch was prepared by the step before."""
        use_next = next_idx is not None and self._ahead_supported(idx.shape[1])
        return (idx.data_ptr(), labels.data_ptr(), tuple(idx.shape), labels.dtype,
                next_idx.data_ptr() if use_next else None, entry["slot"] if entry else None)

    def prepare(self, idx: torch.Tensor, labels: torch.Tensor, next_idx: Optional[torch.Tensor] = None,
                ahead_slot: Optional[int] = None):
        """use_graph: capture the step of a resident batch (step_tensors(idx, labels, resident=True, next_idx=next_idx))
        now, without stepping; the training state is left as it was.  ahead_slot (0 / 1): capture the form the step
        takes when the step before it announced this batch and sorted it into that plan slot (slots alternate from
        step to step, starting with 1 for the step after an unannounced one)."""
        if not self.use_graph:
            return
        entry = None
        if ahead_slot is not None and self._ahead_supported(idx.shape[1]):
            entry = {"key": self._tensor_key(idx), "slot": int(ahead_slot), "B": idx.shape[1],
                     "marked": self.exact and self._ahead_kind == "plan"}
        key = self._graph_key(idx, labels, next_idx, entry)
        if key not in self._graphs:
            saved = self._ahead
            self._graphs[key] = self._capture(idx, labels, next_idx, entry, run_prologue=entry is not None)
            self._ahead = saved

    def release_graphs(self):
        """Drop every captured graph (graphs that captured collectives must go before their communicator does)."""
        self._graphs.clear()
        getattr(self, "_key_graphs", {}).clear()

    def _run_prologue(self, idx, entry):
        """What the step before would have done for `idx` had it been announced (entry: the record it would have left)."""
        p = self.plan(entry["B"], entry["slot"])
        p.sort_columnar(idx)
        if entry["marked"]:
            p.catchup_plan(*self._row_states(), mode=0)

    def _capture(self, idx, labels, next_idx=None, entry=None, run_prologue=False):
        """Capture one step reading `idx`/`labels` in place; the model state is left untouched.  entry: the record of
        work done ahead (self._ahead) the step starts from; run_prologue: that work was not really done -- do it for
        the warm-up steps (it is undone with everything else)."""
        ent = {"idx": idx, "labels": labels, "next_idx": next_idx}
        # snapshot everything a step mutates, warm up (first launches set kernel attributes), restore
        snap = [(t, t.clone()) for t in self._mutable_state()]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), ops.use_workspace(self.ws):
            if run_prologue:
                self._run_prologue(idx, entry)
            for _ in range(2):
                self._ahead = dict(entry) if entry else None
                self._step_impl(idx, labels, next_idx)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for t, c in snap:
            t.copy_(c)
        self._ahead = dict(entry) if entry else None
        self._before_capture()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode=self._capture_error_mode), \
                ops.use_workspace(self.ws):
            ent["loss"] = self._step_impl(idx, labels, next_idx)
        ent["ahead_after"] = self._ahead
        if self._graph_pool is None:
            self._graph_pool = g.pool()
        # the first launch of a graph pays for its instantiation on the device (some 0.1 ms): pay it here
        g.replay()
        torch.cuda.synchronize()
        for t, c in snap:       # keep the state exactly as before
            t.copy_(c)
        self._ahead = dict(entry) if entry else None
        ent["graph"] = g
        return ent

    def _before_capture(self):
        """Between the warm-up steps and the capture (trainers whose warm-up steps issue collectives wait here)."""

    def _mutable_state(self):
        ts = [self.flat_w, self.step_count, self.hyper, self.scal]
        for x in (self.flat_m, self.flat_v):
            if x is not None:
                ts.append(x)
        for t in self.tables:
            ts.append(t["w"])
            for k in ("m", "v"):
                if t[k] is not None:
                    ts.append(t[k])
            if t.get("rs") is not None:
                ts += t["rs"].tensors()
        if self.shrink is not None:
            ts += self.shrink.tensors()
        for b in self.blocks:
            if b.normalization == "bn":
                ts += [b.norm_layer.running_mean, b.norm_layer.running_var]
        return ts

    def step(self, inputs, labels) -> torch.Tensor:
        """BaseModel.train_step signature: dict-of-columns batch and dict of label columns."""
        m = self.model
        x = m.features_to_divice(inputs)
        y = m.labels_to_device(labels)
        idx = m.embedding_layer.index_matrix(x)
        self._bump_bn_counters()
        if y.dtype not in (torch.int64, torch.float32):
            y = y.float()
        return self.step_tensors(idx, y.contiguous())
