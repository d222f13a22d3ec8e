"""BaseModel.train_step (model/utils/BaseModel.py:141-150) for any model of this package, with table rows updated
only where a batch touches them.

The fused engine (engine.FusedTrainer) hand-schedules the whole step of the LR / FM / DeepFM / DNN / WideDeep
family.  The interaction models (FFM, PNN, NFM, AFM, DCN, DeepCrossing) differ in how they wire the embeddings into
their dense part, so here the model's own ``forward`` runs under autograd -- interaction operators and MLP blocks are
HIP kernels behind autograd Functions -- while everything that concerns the tables leaves autograd:

* lookups are captured (functional.TableCapture): they return leaf tensors, and after ``backward()`` their ``.grad``
  is the per-sample gradient of the gathered rows / LR sums / FFM terms;
* those gradients go through the deterministic sort + segmented reduction (K3) or, for FFM's field-aware rows,
  rs_ffm_bwd_rows, which also produce the tables' share of the global gradient norm;
* clip_grad_norm_(max_norm) becomes one device-side coefficient; the dense parameters (one flat buffer) and the
  touched table rows are stepped by the HIP optimizer kernels.

What the reference does instead (and what this replaces): dense (V_f, d) gradients for every table, a norm over all of
them, and a dense optimizer sweep over every row each step (torch.optim on nn.Embedding weights).

Semantics: identical to the reference's dense optimizers for SGD and Adagrad; Adam/AdamW are the "lazy" variant
(rows a batch does not touch keep weights and moments).  ``emb_regular`` (a gradient on every row) is not supported.
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import nn

from . import functional as Fn
from . import ops
from ._lib import RSError

N_DENSE_PARTIALS = 256


class SparseTrainer:
    def __init__(self, model, max_norm: float = 10.0, optimizer_type=None, lr=None):
        self.model = model
        self.max_norm = float(max_norm)
        self.opt = optimizer_type or model.optimizer_type
        if self.opt not in ops.OPT_CODES:
            raise ValueError("Invalid value for optimizer")
        self.lr = float(lr if lr is not None else model.learning_rate)
        if model.emb_regular:
            raise RSError("SparseTrainer: emb_regular puts a gradient on every table row; it has no touched-rows form")
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RSError("SparseTrainer: the model must live on the MI355X (there is no CPU path)")
        self.device = dev
        table_params = set()
        for mod in model.modules():
            if isinstance(mod, nn.Embedding):
                table_params.update(id(p) for p in mod.parameters())
        self.dense = [p for p in model.parameters() if id(p) not in table_params and p.requires_grad]
        self._flatten()
        adam = self.opt in ("Adam", "AdamW")
        self.need_m = adam or self.opt == "Adagrad"
        self.need_v = adam
        self.flat_m = torch.zeros_like(self.flat_w) if self.need_m else None
        self.flat_v = torch.zeros_like(self.flat_w) if self.need_v else None
        self.hyper = ops.make_hyper(dev, self.lr, eps=1e-10 if self.opt == "Adagrad" else 1e-8,
                                    weight_decay=0.01 if self.opt == "AdamW" else 0.0)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.scal = torch.tensor([0.0, 1.0], dtype=torch.float32, device=dev)
        self._state: Dict[int, dict] = {}       # per table holder: optimizer moments
        self._plans: Dict[tuple, object] = {}
        self._partials = None
        self.last = {}

    # ------------------------------------------------------------------------------------------
    def _flatten(self):
        """Dense parameters and their gradients as views into two flat buffers: one norm launch, one optimizer
        launch (every view 16-byte aligned)."""
        offs, o = [], 0
        for p in self.dense:
            offs.append(o)
            o += (p.numel() + 3) // 4 * 4
        self.flat_w = torch.zeros(max(o, 4), dtype=torch.float32, device=self.device)
        self.flat_g = torch.zeros_like(self.flat_w)
        self._gviews = []
        with torch.no_grad():
            for p, off in zip(self.dense, offs):
                v = self.flat_w[off:off + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                self._gviews.append(self.flat_g[off:off + p.numel()].view_as(p))

    def decay_lr(self, factor=0.1, min_lr=1e-6):
        self.lr = max(self.lr * factor, min_lr)
        self.hyper[0:1].fill_(self.lr)
        return self.lr

    def _table_state(self, holder):
        st = self._state.get(id(holder))
        if st is None or st["w"] is not holder.stacked:
            w = holder.stacked
            st = {"w": w, "m": torch.zeros_like(w) if self.need_m else None,
                  "v": torch.zeros_like(w) if self.need_v else None}
            self._state[id(holder)] = st
        return st

    def _plan(self, rec, B):
        holder, kind = rec["holder"], rec["kind"]
        key = (id(holder), kind, B)
        p = self._plans.get(key)
        if p is None:
            if kind == "ffm":
                p = ops.FFMPlan(holder.layout, cap=B, d=holder.dim // holder.layout.num_fields)
            else:
                p = ops.SparsePlan(holder.layout, cap=B, d=holder.dim if kind == "emb" else 1, with_lr=(kind == "lr"))
            self._plans[key] = p
        return p

    # ------------------------------------------------------------------------------------------
    def step(self, inputs, labels) -> torch.Tensor:
        """BaseModel.train_step signature: dict-of-columns batch and dict of label columns; returns the loss."""
        m = self.model
        y = m.labels_to_device(labels)
        self.flat_g.zero_()
        for p, gv in zip(self.dense, self._gviews):
            p.grad = gv                              # autograd accumulates into the flat buffer in place
        with Fn.TableCapture() as cap:
            out = m.forward(inputs)
            loss = m.criterion(out.squeeze(1), y.float())
            if m.net_regular:
                loss = loss + m.regularization_loss()
            loss.backward()
        seen = set()
        jobs = []
        n_total = N_DENSE_PARTIALS
        for rec in cap.records:
            g = rec["out"].grad
            if g is None:
                continue
            if id(rec["holder"]) in seen:
                raise RSError("SparseTrainer: a table was looked up twice in one forward pass")
            seen.add(id(rec["holder"]))
            B = rec["idx"].shape[1]
            plan = self._plan(rec, B)
            n = plan.n_partials_wide if rec["kind"] == "ffm" else plan.n_partials
            jobs.append((rec, plan, g, n))
            n_total += n
        if self._partials is None or self._partials.numel() != n_total:
            self._partials = torch.zeros(n_total, dtype=torch.float32, device=self.device)
        P = self._partials
        off = 0
        for rec, plan, g, n in jobs:
            idx, kind = rec["idx"], rec["kind"]
            B = idx.shape[1]
            if kind == "emb":
                plan.sort_columnar(idx)
                plan.reduce(P[off:off + n], gout=g.contiguous().view(B * plan.F, plan.d))
            elif kind == "lr":
                plan.sort_columnar(idx)
                plan.reduce(P[off:off + n], dlr=g.contiguous().view(B))
            else:
                plan.rows(idx, rec["holder"].stacked, g.contiguous().view(B), P[off:off + n])
            off += n
        ops.sumsq_partials(self.flat_g, P[off:off + N_DENSE_PARTIALS])
        ops.clip_and_prepare(P, self.max_norm, self.scal, self.step_count, self.hyper)
        ops.dense_update(self.flat_w, self.flat_g, self.flat_m, self.flat_v, self.opt, self.hyper, self.scal)
        for rec, plan, g, n in jobs:
            st = self._table_state(rec["holder"])
            if rec["kind"] == "emb":
                plan.update(self.opt, self.hyper, self.scal, table=st["w"], m=st["m"], v=st["v"])
            elif rec["kind"] == "lr":
                plan.update(self.opt, self.hyper, self.scal, table1=st["w"].view(-1),
                            m1=st["m"].view(-1) if st["m"] is not None else None,
                            v1=st["v"].view(-1) if st["v"] is not None else None)
            else:
                plan.update_wide(self.opt, self.hyper, self.scal, st["w"], st["m"], st["v"])
        self.last = {"records": cap.records}
        return loss.detach()
