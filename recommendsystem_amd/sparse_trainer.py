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

Semantics: identical to the reference's dense optimizers for SGD and Adagrad.  Under Adam/AdamW a row that a batch
does not touch still moves in the reference (zero gradient, decaying moments): with ``exact_adam=True`` (default) each
table carries a per-row step counter and a row's skipped steps are replayed right before it is next read
(rs_adam_catchup; ``flush()`` brings every row up to date), which reproduces the dense optimizer;
``exact_adam=False`` is the "lazy" variant (absent rows keep weights and moments).  ``emb_regular`` (a gradient on
every row) is not supported.

``use_graph=True`` replays the whole step, autograd backward included, from a HIP graph (one per batch shape), which
removes the host cost of the ~60 launches and of the autograd bookkeeping.
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
    def __init__(self, model, max_norm: float = 10.0, optimizer_type=None, lr=None, use_graph: bool = False,
                 exact_adam: bool = True):
        self.model = model
        self.use_graph = bool(use_graph)
        self._graphs = {}
        self._graph_pool = None
        self._capture_stream = None
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
        self.exact = bool(exact_adam) and adam
        self._bctab = None
        self.flat_m = torch.zeros_like(self.flat_w) if self.need_m else None
        self.flat_v = torch.zeros_like(self.flat_w) if self.need_v else None
        self.hyper = ops.make_hyper(dev, self.lr, eps=1e-10 if self.opt == "Adagrad" else 1e-8,
                                    weight_decay=0.01 if self.opt == "AdamW" else 0.0)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.scal = torch.tensor([0.0, 1.0], dtype=torch.float32, device=dev)
        self._state: Dict[int, dict] = {}       # per table holder: optimizer moments
        self._plans: Dict[tuple, object] = {}
        self._partials = None
        self._seeds = {}             # gradient scale -> the tensor handed to backward() as the loss's incoming gradient
        self.last = {}
        # scratch of this trainer's launches (split-K slabs, column-sum partials): private, so that a captured step
        # never shares a buffer with another model or with module-path calls on the same device
        self.ws = ops.Workspace(dev)
        # Learned from the previous step: which plan sorts the batch a table is looked up with, and which tables share
        # it.  The first lookup of such a group then sorts the batch and replays the skipped steps of every table of
        # the group in one launch (the sort keeps only F workgroups busy) instead of a catch-up launch per table now
        # and a sort launch in the backward pass.
        self._group_of: Dict[int, dict] = {}
        self._sorted: Dict[int, torch.Tensor] = {}      # per step: id(plan) -> the index matrix it has sorted
        self.check_every = 256           # steps between reads of the device-side error flags (a host synchronisation)
        self._unchecked = 0

    # ------------------------------------------------------------------------------------------
    def _flatten(self):
        """Dense parameters and their gradients as views into two flat buffers: one norm launch, one optimizer
        launch (every view 16-byte aligned)."""
        offs, o = [], 0
        for p in self.dense:
            offs.append(o)
            o += (p.numel() + 3) // 4 * 4
        self.flat_w = torch.zeros(max(o, 4), dtype=torch.float32, device=self.device)
        # four spare floats behind the gradients: the multi-GPU trainer sends its scalar reductions along with the
        # one all-reduce of this buffer
        self.flat_g_ext = torch.zeros(self.flat_w.numel() + 4, dtype=torch.float32, device=self.device)
        self.flat_g = self.flat_g_ext[:self.flat_w.numel()]
        self._gviews = []
        with torch.no_grad():
            for p, off in zip(self.dense, offs):
                v = self.flat_w[off:off + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                self._gviews.append(self.flat_g[off:off + p.numel()].view_as(p))

    def decay_lr(self, factor=0.1, min_lr=1e-6):
        self.flush()                 # rows left behind are replayed with the rate their skipped steps had
        self.lr = max(self.lr * factor, min_lr)
        self.hyper[0:1].fill_(self.lr)
        return self.lr

    def flush(self):
        """Bring every table row up to the current step (only the exact Adam mode leaves rows behind)."""
        for st in self._state.values():
            if st.get("rs") is not None:
                st["rs"].flush()

    def _table_state(self, holder):
        st = self._state.get(id(holder))
        if st is not None and st["w"] is not holder.stacked:
            raise RSError("SparseTrainer: a table was moved to other storage after training began "
                          "(model.to(another device)?); build a new trainer")
        if st is None:
            w = holder.stacked
            st = {"w": w, "m": torch.zeros_like(w) if self.need_m else None,
                  "v": torch.zeros_like(w) if self.need_v else None, "rs": None}
            holder.flush_rows()          # an earlier trainer's pending steps, then its counters go
            holder.row_state = None
            if self.exact:
                st["rs"] = holder.row_state = ops.RowAdam(w, st["m"], st["v"], self.opt, self.hyper, self.step_count,
                                                          bctab=self._bctab)
                self._bctab = st["rs"].bctab
            self._state[id(holder)] = st
        return st

    def _on_lookup(self, holder, idx):
        """A captured lookup is about to read `holder`'s table in a training step: replay the skipped steps of the
        rows it reads."""
        rs = self._table_state(holder)["rs"]
        if rs is None:
            return
        ent = self._group_of.get(id(holder))
        if ent is not None and ent["B"] == idx.shape[1]:
            plan = ent["plan"]
            done = self._sorted.get(id(plan))
            if done is idx:
                return                                # caught up together with the sort of this batch
            if done is None:
                rs_of = lambda h: self._table_state(h)["rs"] if h is not None else None
                plan.sort_and_catchup(idx, rs_of(ent["main"]), rs_of(ent["lr"]))
                self._sorted[id(plan)] = idx
                return
        rs.catch_up(idx, holder.layout, 0)            # the update of this step stamps the rows it steps

    def _plan(self, group, B):
        """One sort plan per group of lookups that share a batch (same index matrix and field vocabularies)."""
        key = (tuple(sorted((id(r["holder"]), r["kind"]) for r in group.values())), B)
        p = self._plans.get(key)
        if p is None:
            any_rec = next(iter(group.values()))
            layout = any_rec["holder"].layout
            if "ffm" in group:
                h = group["ffm"]["holder"]
                p = ops.FFMPlan(layout, cap=B, d=h.dim // layout.num_fields, with_lr="lr" in group)
            else:
                p = ops.SparsePlan(layout, cap=B, d=group["emb"]["holder"].dim if "emb" in group else 1,
                                   with_lr="lr" in group)
            self._plans[key] = p
        return p

    @staticmethod
    def _groups(records):
        """Records with a gradient, grouped by shared index matrix: at most one lookup of each kind per group, and
        an FFM lookup does not share a plan with an embedding lookup (different chunk windows)."""
        groups, seen = [], set()
        for rec in records:
            if rec["out"].grad is None:
                continue
            if id(rec["holder"]) in seen:
                raise RSError("SparseTrainer: a table was looked up twice in one forward pass")
            seen.add(id(rec["holder"]))
            for g in groups:
                first = next(iter(g.values()))
                if (first["idx"] is rec["idx"] and rec["kind"] not in g
                        and first["holder"].layout.vocab_sizes == rec["holder"].layout.vocab_sizes
                        and not ({"emb", "ffm"} <= set(g) | {rec["kind"]})):
                    g[rec["kind"]] = rec
                    break
            else:
                groups.append({rec["kind"]: rec})
        return groups

    def step_tensors(self, idx: torch.Tensor, labels: torch.Tensor, resident: bool = False) -> torch.Tensor:
        """One step on a device-resident columnar batch (idx [F, B] int64 in the embedding layer's column order,
        labels [B]), the calling convention of the fused engine.  With use_graph the whole step -- forward, autograd
        backward, touched-rows reduction, optimizer -- is replayed from a HIP graph; resident=True promises that these
        exact tensors stay alive and are re-used (one graph per buffer pair, no copies), otherwise the batch is copied
        into the graph's static input buffers first."""
        self._tick()
        if not self.use_graph:
            with ops.use_workspace(self.ws):
                return self._step_impl(self._columns(idx), labels)
        if resident:
            key = (idx.data_ptr(), labels.data_ptr(), tuple(idx.shape), labels.dtype)
            ent = self._graphs.get(key)
            if ent is None:
                ent = self._graphs[key] = self._capture(idx, labels)
        else:
            key = ("static", tuple(idx.shape), labels.dtype)
            ent = self._graphs.get(key)
            if ent is None:
                ent = self._graphs[key] = self._capture(idx.clone(), labels.clone())
            ent["idx"].copy_(idx)
            ent["labels"].copy_(labels)
        ent["graph"].replay()
        return ent["loss"]

    def prepare(self, idx: torch.Tensor, labels: torch.Tensor):
        """use_graph: capture the step of a resident batch (step_tensors(idx, labels, resident=True)) now, without
        stepping; model and optimizer state are left as they were."""
        if not self.use_graph:
            return
        key = (idx.data_ptr(), labels.data_ptr(), tuple(idx.shape), labels.dtype)
        if key not in self._graphs:
            self._graphs[key] = self._capture(idx, labels)

    def _tick(self):
        self._unchecked += 1
        if self.check_every and self._unchecked >= self.check_every:
            self.check()

    def check(self):
        """Raise IndexError if a lookup saw an out-of-range index since the last check (device sync)."""
        self._unchecked = 0
        for mod in self.model.modules():
            for name in ("_tab", "_tab1"):
                holder = getattr(mod, name, None)
                if holder is not None and getattr(holder, "layout", None) is not None:
                    holder.layout.check_indices()

    def _columns(self, idx):
        return {c: idx[i] for i, c in enumerate(self.model.embedding_layer.lookup_columns)}

    def _mutable_state(self):
        ts = [self.flat_w, self.step_count, self.hyper, self.scal]
        ts += [x for x in (self.flat_m, self.flat_v) if x is not None]
        dense = {id(p) for p in self.dense}
        ts += [p.data for p in self.model.parameters() if id(p) not in dense]
        ts += list(self.model.buffers())
        for st in self._state.values():
            ts += [st[k] for k in ("m", "v") if st[k] is not None]
            if st["rs"] is not None:
                ts += st["rs"].tensors()
        return ts

    def _capture(self, idx, labels):
        """Capture one step reading `idx`/`labels` in place; model and optimizer state are left as they were."""
        known = set(self._state)
        snap = [(t, t.clone()) for t in self._mutable_state()]
        x = self._columns(idx)
        # warm-up and capture on one stream: autograd's accumulation nodes remember the stream of their first use
        if self._capture_stream is None:
            self._capture_stream = torch.cuda.Stream()
        side = self._capture_stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.use_workspace(self.ws):
            for _ in range(2):          # first launches set kernel attributes, size workspaces and plans
                self._step_impl(x, labels)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        def restore():
            for t, c in snap:
                t.copy_(c)
            for k, st in self._state.items():
                if k not in known:          # moments created by the warm-up steps start from zero
                    for name in ("m", "v"):
                        if st[name] is not None:
                            st[name].zero_()
                    if st["rs"] is not None:
                        st["rs"].last.zero_()
        restore()
        g = torch.cuda.CUDAGraph()
        # no cycle collection while the stream is capturing: a collected tensor from outside the graph's pool would
        # make the allocator record events, which a capturing stream refuses
        import gc
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(g, pool=self._graph_pool, stream=side), ops.use_workspace(self.ws):
                loss = self._step_impl(x, labels)
        finally:
            if was_enabled:
                gc.enable()
        if self._graph_pool is None:
            self._graph_pool = g.pool()
        torch.cuda.synchronize()
        restore()
        return {"idx": idx, "labels": labels, "graph": g, "loss": loss}

    # ------------------------------------------------------------------------------------------
    def step(self, inputs, labels) -> torch.Tensor:
        """BaseModel.train_step signature: dict-of-columns batch and dict of label columns; returns the loss."""
        if self.use_graph:
            m = self.model
            x = m.features_to_divice(inputs)
            y = m.labels_to_device(labels)
            if y.dtype not in (torch.int64, torch.float32):
                y = y.float()
            return self.step_tensors(m.embedding_layer.index_matrix(x), y.contiguous())
        self._tick()
        with ops.use_workspace(self.ws):
            return self._step_impl(inputs, self.model.labels_to_device(labels))

    def _forward_backward(self, inputs, y, provider=None, grad_scale: float = 1.0, defer_finishes: bool = False):
        """The model's forward and autograd backward with captured table lookups; dense gradients land in flat_g.
        defer_finishes: the split-K and column-sum finishing launches of the pass are left to the caller
        (self._pending_finishes = (SplitKParts or None, ColsumParts or None)), who lets them ride in another launch."""
        m = self.model
        self._sorted.clear()
        self.flat_g_ext.zero_()
        for p, gv in zip(self.dense, self._gviews):
            p.grad = gv                              # autograd accumulates into the flat buffer in place
        # MLP blocks write weight / bias gradients straight into the flat buffer (no autograd accumulation launch per
        # parameter) and leave their split-K slabs to one reducing launch; with an L2 term autograd also adds into
        # those views, so the plain route is kept there
        direct = not m.net_regular
        sinks = {id(p): gv for p, gv in zip(self.dense, self._gviews)} if direct else None
        splitk = ops.SplitKParts(ops.default_workspace(self.device)) if direct else None
        colsum = ops.ColsumParts(ops.default_workspace(self.device), tag="_pass") if direct else None
        with Fn.TableCapture(provider, grad_sinks=sinks, splitk=splitk, colsum=colsum,
                             on_lookup=self._on_lookup if self.exact else None) as cap:
            cap.rng_step = self.step_count          # numbers the dropout masks: fresh ones at every (replayed) step
            if hasattr(m, "logits") and m.criterion is torch.nn.functional.binary_cross_entropy:
                # sigmoid + BCE + dL/dlogit in one launch instead of torch's five
                yl = y if y.dtype in (torch.float32, torch.int64) else y.float()
                loss = Fn.bce_of_logits(m.logits(inputs), yl.contiguous(), assume_grad=grad_scale)
            else:
                out = m.forward(inputs)
                loss = m.criterion(out.squeeze(1), y.float())
            if m.net_regular:
                loss = loss + m.regularization_loss()
            # the gradient arriving at the loss comes from a tensor kept for the purpose: backward() would otherwise
            # fill a fresh one (and a scaled loss multiply it) in launches of their own
            seed = self._seeds.get(float(grad_scale))
            if seed is None or seed.shape != loss.shape:
                seed = self._seeds[float(grad_scale)] = torch.full_like(loss, float(grad_scale))
            loss.backward(seed)
            self._pending_finishes = (None, None)
            if defer_finishes:
                self._pending_finishes = (splitk, colsum)
            else:
                if splitk is not None:
                    splitk.finish()
                if colsum is not None:
                    colsum.finish()      # every bias gradient (and other column sum) of the pass, one launch per eight
        return cap, loss

    def _finish_pending(self):
        splitk, colsum = self._pending_finishes
        self._pending_finishes = (None, None)
        if splitk is not None:
            splitk.finish()
        if colsum is not None:
            colsum.finish()

    def _step_impl(self, inputs, y) -> torch.Tensor:
        cap, loss = self._forward_backward(inputs, y, defer_finishes=True)
        if self.exact and any(r["out"].grad is None for r in cap.records):
            self._finish_pending()
            raise RSError("SparseTrainer: a table was looked up but its rows do not reach the loss; with exact Adam "
                          "every looked-up row must be stepped (drop the lookup or pass exact_adam=False)")
        jobs = []
        n_total = N_DENSE_PARTIALS
        for group in self._groups(cap.records):
            B = next(iter(group.values()))["idx"].shape[1]
            plan = self._plan(group, B)
            n = plan.n_partials + (plan.n_partials_wide if "ffm" in group else 0)
            jobs.append((group, plan, n))
            n_total += n
        if self._partials is None or self._partials.numel() != n_total:
            self._partials = torch.zeros(n_total, dtype=torch.float32, device=self.device)
        P = self._partials
        off = 0
        # one looked-up table set (every model but FFM): the dense gradient's norm partials come out of the embedding
        # reduction's second launch and the flat dense step shares the touched-rows update's launch, as in the fused engine
        # (an FFM model's 1-dim LR lookup shares the batch with the field-aware one: its reduction and update carry them)
        single = len(jobs) == 1 and ("ffm" not in jobs[0][0] or "lr" in jobs[0][0])
        # ... and the finishing additions of the dense backward (weight-gradient slabs, bias-gradient partial sums) ride
        # in the reduction's first launch; otherwise they get launches of their own now
        carry = (None, None)
        if single:
            splitk, colsum = self._pending_finishes
            self._pending_finishes = (None, None)
            carry = (colsum.hand_over() if colsum is not None else None, splitk.hand_over() if splitk is not None else None)
        else:
            self._finish_pending()
        for group, plan, n in jobs:
            idx = next(iter(group.values()))["idx"]
            B = idx.shape[1]
            grad = lambda kind: group[kind]["out"].grad.contiguous() if kind in group else None
            if "ffm" in group:
                nw = plan.n_partials_wide
                plan.rows(idx, group["ffm"]["holder"].stacked, grad("ffm").view(B), P[off:off + nw],
                          sorted_already=self._sorted.get(id(plan)) is idx)
                if "lr" in group:
                    plan.reduce(P[off + nw:off + n], dlr=grad("lr").view(B),
                                dense=self.flat_g if single else None,
                                dense_partials=P[off + n:off + n + N_DENSE_PARTIALS] if single else None,
                                colsum=carry[0], splitk=carry[1])
                else:
                    P[off + nw:off + n].zero_()
            else:
                if self._sorted.get(id(plan)) is not idx:
                    plan.sort_columnar(idx)
                ge, gl = grad("emb"), grad("lr")
                plan.reduce(P[off:off + n], gout=ge.view(B * plan.F, plan.d) if ge is not None else None,
                            dlr=gl.view(B) if gl is not None else None,
                            dense=self.flat_g if single else None,
                            dense_partials=P[off + n:off + n + N_DENSE_PARTIALS] if single else None,
                            colsum=carry[0], splitk=carry[1])
            if self.exact:
                main = group.get("emb") or group.get("ffm")
                for r in group.values():
                    self._group_of[id(r["holder"])] = {"plan": plan, "B": B, "main": main["holder"] if main else None,
                                                       "lr": group["lr"]["holder"] if "lr" in group else None}
            off += n
        if not single:
            ops.sumsq_partials(self.flat_g, P[off:off + N_DENSE_PARTIALS])
        ops.clip_and_prepare(P, self.max_norm, self.scal, self.step_count, self.hyper)
        if not single:
            ops.dense_update(self.flat_w, self.flat_g, self.flat_m, self.flat_v, self.opt, self.hyper, self.scal)
        for group, plan, n in jobs:
            st = {k: self._table_state(r["holder"]) for k, r in group.items()}
            if "ffm" in group:
                plan.update_wide(self.opt, self.hyper, self.scal, st["ffm"]["w"], st["ffm"]["m"], st["ffm"]["v"],
                                 rs=st["ffm"]["rs"])
            e, l = st.get("emb"), st.get("lr")
            if e is not None or l is not None:
                flat = lambda t: t.view(-1) if t is not None else None
                kw = dict(table=e["w"] if e else None, m=e["m"] if e else None, v=e["v"] if e else None,
                          table1=flat(l["w"]) if l else None, m1=flat(l["m"]) if l else None,
                          v1=flat(l["v"]) if l else None, rs=e["rs"] if e else None, rs1=l["rs"] if l else None)
                if single:
                    plan.update_with_dense(self.opt, self.hyper, self.scal, self.flat_w, self.flat_g, self.flat_m,
                                           self.flat_v, **kw)
                else:
                    plan.update(self.opt, self.hyper, self.scal, **kw)
        self.last = {"records": cap.records, "direct_grads": len(cap._sunk)}
        return loss.detach()
