"""Multi-GPU training of the hot path on one 8xMI355X node: one process per GPU.

* batch: data-parallel, every rank steps on its own B samples (weak scaling);
* dense towers: replicated; ONE all-reduce of the flat gradient buffer per step (RCCL over xGMI);
* embedding tables: row-sharded, row i of every field on rank ``i % N`` at local index ``i // N + 1``
  (local row 0 is a zero dummy so "index 0 = padding" holds on every rank; ``% N`` also spreads the
  frequency-sorted hot low rows evenly).  Per step: all-to-all of DISTINCT indices, all-to-all of their
  rows back, and on backward all-to-all of one gradient per distinct index; the owner merges the N
  contributions per row with the same sort + segmented-reduce kernels as the single-GPU path and
  updates its shard.  Buckets have a fixed capacity, so every collective is rectangular and no count
  visits the host;
* gradient clipping: one scalar all-reduce of the sharded ||g||^2.

The reference has no collective of any kind; semantics are those of its single-device step
(model/utils/BaseModel.py:141-150) on the union of all ranks' batches, with mean loss over N*B samples.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from . import ops
from ._lib import RSError, call, ptr, stream
from .engine import FusedTrainer, N_DENSE_PARTIALS
from .sparse_trainer import SparseTrainer
from .ops import FieldLayout, SparsePlan, WidePlan


# -------------------------------------------------------------------------------------------------
# sharding arithmetic (host)
# -------------------------------------------------------------------------------------------------
class ShardSpec:
    def __init__(self, vocab_sizes: List[int], world: int, rank: int):
        self.vocab = [int(v) for v in vocab_sizes]
        self.N, self.rank = int(world), int(rank)
        # rows i < V with i % N == rank, plus the dummy row 0
        self.local_vocab = [self.owned_count(v, self.rank) + 1 for v in self.vocab]

    def owned_count(self, v: int, r: int) -> int:
        return (v - r + self.N - 1) // self.N if v > r else 0

    @staticmethod
    def owner(i, N):
        return i % N

    @staticmethod
    def local_index(i, N):
        return i // N + 1

    def bucket_capacity(self, B: int, slack: float = 1.5, extra: int = 64) -> int:
        """Entries per (owner, field) bucket: distinct keys of one field spread ~uniformly over owners."""
        cap = int(-(-B // self.N) * slack) + extra
        return min(cap, B)

    def buckets(self, B: int, slack: float = 1.5, extra: int = 64, device=None) -> "BucketLayout":
        """Per-field bucket capacities: the statistical bound above, but never more than the rows of the field an owner
        holds (a 100-row field needs 13 slots per owner at N = 8, not B/N * 1.5 + 64) -- those buckets cannot overflow,
        and the all-to-alls carry a third of the bytes on the Criteo_x1 vocabularies."""
        cap = self.bucket_capacity(B, slack, extra)
        return BucketLayout([max(1, min(cap, -(-v // self.N))) for v in self.vocab], device)

    def shard_table(self, full: torch.Tensor) -> torch.Tensor:
        """[sum V_f, d] stacked global table -> this rank's stacked local table [sum local_vocab, d]."""
        out, o = [], 0
        for v in self.vocab:
            rows = full[o + self.rank:o + v:self.N]
            out.append(torch.zeros((1,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device))
            out.append(rows)
            o += v
        return torch.cat(out, 0).contiguous()

    def unshard_into(self, full: torch.Tensor, shards: List[torch.Tensor]):
        """Inverse of shard_table given every rank's local table (rank order)."""
        for r, sh in enumerate(shards):
            o = lo = 0
            for v in self.vocab:
                n = self.owned_count(v, r)
                full[o + r:o + v:self.N] = sh[lo + 1:lo + 1 + n]
                o += v
                lo += n + 1


class BucketLayout:
    """Where bucket (owner o, field f) lies in the buffers of the exchange: capf[f] slots per owner, boff[f] = the slots
    of the fields before f, S slots per owner in all.  Key buffers are [N, S + F] int64 (count slot, then keys), row and
    gradient buffers [N, S, d]."""

    def __init__(self, capf: List[int], device=None):
        self.capf = [int(c) for c in capf]
        self.boff = [0]
        for c in self.capf[:-1]:
            self.boff.append(self.boff[-1] + c)
        self.S = sum(self.capf)
        self.cap_max = max(self.capf)
        self.F = len(self.capf)
        self.capf_t = torch.tensor(self.capf, dtype=torch.int32, device=device)
        self.boff_t = torch.tensor(self.boff, dtype=torch.int32, device=device)

    def list_tables(self, N: int, device=None):
        """(sub_off, pay_base) of the N*F key lists an owner receives, list l = s*F + f: first key of bucket (s, f) in the
        flattened key buffer, and the position of its first row in the row / gradient buffers."""
        s = torch.arange(N, dtype=torch.int64).view(N, 1)
        boff = torch.tensor(self.boff, dtype=torch.int64).view(1, self.F)
        f = torch.arange(self.F, dtype=torch.int64).view(1, self.F)
        sub_off = (s * (self.S + self.F) + boff + f + 1).reshape(-1).to(torch.int32)
        pay_base = (s * self.S + boff).reshape(-1).to(torch.int32)
        return sub_off.to(device), pay_base.to(device)


# -------------------------------------------------------------------------------------------------
# collectives
# -------------------------------------------------------------------------------------------------
class Comm:
    """torch.distributed wrapper.  stage_cpu=True moves tensors through host memory around each
    collective (for the gloo backend, which is what the CPU tests and single-GPU rehearsals use).
    Collectives are issued on torch's current stream, like every kernel of the step."""

    def __init__(self, group=None, stage_cpu: bool = False):
        self.group = group
        self.stage_cpu = stage_cpu
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def second(self) -> "Comm":
        """A communicator of its own over the same ranks: collectives issued through it on another stream run beside
        those of this one (the dense all-reduce beside the gradient all-to-all).  Collective call: every rank."""
        ranks = dist.get_process_group_ranks(self.group) if self.group is not None else None
        return Comm(dist.new_group(ranks=ranks), stage_cpu=self.stage_cpu)

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor):
        """Equal split: chunk s of `inp` goes to rank s; chunk s of `out` comes from rank s."""
        if self.stage_cpu and inp.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o.view(-1), inp.cpu().view(-1), group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out.view(-1), inp.view(-1), group=self.group)

    def all_reduce(self, t: torch.Tensor):
        if self.stage_cpu and t.is_cuda:
            c = t.cpu()
            dist.all_reduce(c, group=self.group)
            t.copy_(c)
        else:
            dist.all_reduce(t, group=self.group)

    def broadcast(self, t: torch.Tensor, src: int = 0):
        if self.stage_cpu and t.is_cuda:
            c = t.cpu()
            dist.broadcast(c, src, group=self.group)
            t.copy_(c)
        else:
            dist.broadcast(t, src, group=self.group)

    def all_gather_into(self, out: torch.Tensor, inp: torch.Tensor):
        """out [world, *inp.shape] (persistent) <- every rank's `inp`, in rank order."""
        if self.stage_cpu and inp.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather(list(o.unbind(0)), inp.cpu().contiguous(), group=self.group)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out.view(-1), inp.contiguous().view(-1), group=self.group)

    def all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        src = t.cpu() if (self.stage_cpu and t.is_cuda) else t
        outs = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(outs, src.contiguous(), group=self.group)
        return [o.to(t.device) for o in outs]


# -------------------------------------------------------------------------------------------------
# device kernels of the exchange (HIP)
# -------------------------------------------------------------------------------------------------
def payload_width(d: int, with_emb: bool, with_lr: bool):
    """(floats per position of the row / gradient buffers, column of the 1-dim value): the d-wide row first; the value
    of the 1-dim (LR) table travels in a spare column behind it -- four floats wide, so that positions stay 16-byte
    aligned -- instead of in a buffer and an all-to-all of its own."""
    if with_emb:
        return (d + 4, d) if with_lr else (d, 0)
    return 1, 0


class HipShardKernels:
    """The device side of the exchange on librs_hip.so.  tests/test_dist_cpu.py substitutes a torch-CPU
    stand-in with the same methods to rehearse the protocol under gloo."""

    def __init__(self, spec: ShardSpec, B: int, d: int, with_emb: bool, with_lr: bool, buckets: BucketLayout, device):
        self.spec, self.B, self.d, self.bk = spec, B, d, buckets
        self.with_emb, self.with_lr = with_emb, with_lr
        self.W, self.lr_col = payload_width(d, with_emb, with_lr)
        self.F = len(spec.vocab)
        self.layout_req = FieldLayout(spec.vocab, device)            # requester: global indices
        self.layout_own = FieldLayout(spec.local_vocab, device)      # owner: local indices (+1 shift)
        self.plan_req = SparsePlan(self.layout_req, cap=B, d=d, with_lr=with_lr)
        self.plan_own = SparsePlan(self.layout_own, cap=spec.N * buckets.cap_max, d=d, with_lr=with_lr)
        N, F = spec.N, self.F
        self.sub_off, self.pay_base = buckets.list_tables(N, device)
        self.sub_cnt = torch.zeros(N * F, dtype=torch.int32, device=device)
        self.scratch_partials = torch.zeros(self.plan_req.n_partials, dtype=torch.float32, device=device)
        # where the row of entry (f, b) lies among the received rows (position N*S: the row of zeros)
        self.slot_map = torch.full((F, B), N * buckets.S, dtype=torch.int32, device=device)

    def _cols(self, buf):
        """(d-wide rows, 1-dim values) of a [*, W] row or gradient buffer, as the pointers the kernels take."""
        flat = buf.view(-1)
        return (flat if self.with_emb else None), (flat[self.lr_col:] if self.with_lr else None)

    def sort_batch(self, idx):
        self.plan_req.sort_columnar(idx)

    def route(self, send_idx, route, overflow):
        bk = self.bk
        call("rs_shard_route", ptr(self.plan_req.plan), self.F, self.B, ptr(self.layout_req.row_offset_t), self.spec.N,
             ptr(bk.capf_t), ptr(bk.boff_t), bk.S, ptr(send_idx), ptr(route), ptr(overflow), stream())
        self._route = route

    def counts(self, recv_idx):
        """Sub-list lengths of the received buckets (once per step, for the catch-up and the merge) and, in the same
        launch, the requester's slot map."""
        bk = self.bk
        call("rs_shard_counts", ptr(recv_idx), self.spec.N, self.F, ptr(bk.capf_t), ptr(bk.boff_t), bk.S,
             ptr(self.sub_cnt), ptr(self.plan_req.plan), ptr(self._route), self.B,
             ptr(self.slot_map) if self.slot_map is not None else None, self.B, self.F, stream())

    def gather(self, recv_idx, table, table1, pay):
        bk = self.bk
        rows, rows1 = self._cols(pay)
        call("rs_shard_gather", ptr(recv_idx), ptr(self.layout_own.row_offset_t), ptr(self.layout_own.vocab_t),
             self.spec.N, self.F, ptr(bk.capf_t), ptr(bk.boff_t), bk.S, bk.cap_max, self.d, ptr(table), ptr(table1),
             ptr(rows), ptr(rows1), self.W, self.W, ptr(self.layout_own.err_flag), stream())

    def merge_plan(self, recv_idx):
        """Owner: the plan of the received key lists -- every distinct row once, with the entries that name it.  It
        depends on the keys alone, so it is made as soon as they arrive: the replay of skipped Adam steps then visits
        every row once, and the backward pass finds its segments ready."""
        N = self.spec.N
        # every received bucket is ascending (rs_shard_route keeps key order): a merge, not a sort
        self.plan_own.merge_lists(recv_idx.view(-1), self.sub_off, self.sub_cnt, self.pay_base, nsrc=N,
                                  max_entries=N * self.bk.cap_max)

    serves_in_catch_up = True     # catch_up(..., pay=...) also writes the rows into the send buffer: no gather() launch

    def catch_up(self, recv_idx, rs, rs1, pay=None):
        """Owner, exact Adam on touched rows: the rows the received buckets name are brought up to the current step
        before they are served (rs / rs1: ops.RowAdam of the shard / the 1-dim shard); once per distinct row.  pay: the
        row send buffer -- every row then goes from the replay's registers to the positions that asked for it."""
        if pay is not None:
            self.plan_own.catchup_serve(rs, rs1, pay.view(-1), self.W, 0, self.lr_col)
        else:
            self.plan_own.catchup_plan(rs, rs1, mode=2)

    def lookup(self, pay, bias, want_emb=True, want_sum=False, want_lr=False, want_fm=False):
        """The requester's fused lookup in the received rows: dict(emb [B,F,d], sum [B,d], lr [B], fm [B]) as K1 leaves
        them on one GPU (ops.gather_lr_fm_fwd), entry (f, b) reading the row of its distinct key."""
        B, F, d = self.B, self.F, self.d
        rows, rows1 = self._cols(pay)
        dev = pay.device
        out = {}
        if want_emb:
            out["emb"] = torch.empty((B, F, d), dtype=torch.float32, device=dev)
        if want_sum:
            out["sum"] = torch.empty((B, d), dtype=torch.float32, device=dev)
        if want_lr:
            out["lr"] = torch.empty((B,), dtype=torch.float32, device=dev)
        if want_fm:
            out["fm"] = torch.empty((B,), dtype=torch.float32, device=dev)
        call("rs_shard_lookup", ptr(self.slot_map), ptr(rows), self.W, ptr(rows1), self.W, ptr(bias), ptr(out.get("emb")),
             ptr(out.get("sum")), None, ptr(out.get("lr")), ptr(out.get("fm")), B, F, d, stream())
        return out

    def expand(self, route, pay, emb, lr1):
        """Per-entry copies of the received rows (emb [B*F, d]) / 1-dim values (lr1 [B*F])."""
        if emb is not None and lr1 is not None:
            raise RSError("shard exchange: rows and 1-dim values in one buffer are read through lookup()")
        if emb is not None:
            rows, _ = self._cols(pay)
            call("rs_shard_lookup", ptr(self.slot_map), ptr(rows), self.W, None, 1, None, ptr(emb), None, None, None, None,
                 self.B, self.F, self.d, stream())
        else:
            call("rs_shard_expand", ptr(self.plan_req.plan), ptr(route), self.F, self.B, self.d, None, ptr(pay), None,
                 ptr(lr1), stream())

    packs_in_reduce = True       # reduce_local(route=..., gbuf=...) writes the send buffer itself: no pack() launch

    def reduce_local(self, gout, dfm, S, emb, dlr, colsum=None, splitk=None, route=None, gbuf=None):
        """One gradient per distinct key of the local batch (plan_req.rowgrad / rowgrad1; with route / gbuf straight into
        the key's bucket position of the gradient send buffer).  colsum / splitk: pending column-sum and split-K finishes
        of the dense backward that ride in the first launch (SparsePlan.reduce)."""
        routed = None
        if route is not None:
            grads, grads1 = self._cols(gbuf)
            routed = (route, grads, grads1, self.W, self.W)
        self.plan_req.reduce(self.scratch_partials, gout=gout, dfm=dfm, S=S, emb=emb, dlr=dlr, Fdiv=self.F,
                             colsum=colsum, splitk=splitk, routed=routed)

    def pack(self, route, gbuf):
        grads, grads1 = self._cols(gbuf)
        call("rs_shard_pack", ptr(self.plan_req.plan), ptr(route), self.F, self.B, self.d,
             ptr(self.plan_req.rowgrad) if grads is not None else None,
             ptr(self.plan_req.rowgrad1) if grads1 is not None else None, ptr(grads), ptr(grads1), self.W, self.W, stream())

    def merge(self, recv_idx, gbuf, partials, dense=None, dense_partials=None):
        """Owner: sum the N contributions per owned row; ||g||^2 partials of the owned rows.  dense / dense_partials:
        the sum-of-squares partials of that flat buffer come out of the reduction's second launch as well."""
        grads, grads1 = self._cols(gbuf)
        self.plan_own.reduce(partials, gout=grads, g1src=grads1, Fdiv=1, ld_gout=self.W if grads is not None else 0,
                             ld_g1=self.W if grads1 is not None else 0, dense=dense, dense_partials=dense_partials)

    def update(self, opt, hyper, scal, table, m, v, table1, m1, v1, dense=None, rs=None, rs1=None):
        """Touched-rows step on the shard; dense = (w, g, m, v): the flat dense step in the same launch."""
        if dense is not None:
            self.plan_own.update_with_dense(opt, hyper, scal, *dense, table=table, m=m, v=v, table1=table1, m1=m1, v1=v1,
                                            rs=rs, rs1=rs1)
        else:
            self.plan_own.update(opt, hyper, scal, table=table, m=m, v=v, table1=table1, m1=m1, v1=v1, rs=rs, rs1=rs1)

    @property
    def n_own_partials(self):
        return self.plan_own.n_partials


class HipShardKernelsWide(HipShardKernels):
    """The same device side for rows of W floats (field-aware FFM rows: W = F*d up to 1024), where a row takes a whole
    workgroup instead of a lane group: wide-row segmented reductions on both the requester and the owner side."""

    def __init__(self, spec: ShardSpec, B: int, W: int, buckets: BucketLayout, device):
        self.spec, self.B, self.d, self.bk = spec, B, W, buckets
        self.with_emb, self.with_lr = True, False
        self.W, self.lr_col = W, 0
        self.F = len(spec.vocab)
        self.layout_req = FieldLayout(spec.vocab, device)
        self.layout_own = FieldLayout(spec.local_vocab, device)
        self.plan_req = WidePlan(self.layout_req, cap=B, W=W)
        self.plan_own = WidePlan(self.layout_own, cap=spec.N * buckets.cap_max, W=W)
        N, F = spec.N, self.F
        self.sub_off, self.pay_base = buckets.list_tables(N, device)
        self.sub_cnt = torch.zeros(N * F, dtype=torch.int32, device=device)
        self.scratch_partials = torch.zeros(self.plan_req.n_partials_wide, dtype=torch.float32, device=device)
        self.slot_map = None

    def route(self, send_idx, route, overflow):
        bk = self.bk
        call("rs_shard_route", ptr(self.plan_req.plan), self.F, self.B, ptr(self.layout_req.row_offset_t), self.spec.N,
             ptr(bk.capf_t), ptr(bk.boff_t), bk.S, ptr(send_idx), ptr(route), ptr(overflow), stream())
        self._route = route

    def lookup(self, *a, **k):
        raise RSError("wide rows are expanded, not looked up")

    def expand(self, route, pay, emb, lr1):
        call("rs_shard_expand", ptr(self.plan_req.plan), ptr(route), self.F, self.B, self.d, ptr(pay), None, ptr(emb), None,
             stream())

    packs_in_reduce = False
    serves_in_catch_up = False

    def catch_up(self, recv_idx, rs, rs1):
        self.plan_own.catchup_plan(rs, rs1, mode=2)

    def reduce_local(self, gout, dfm, S, emb, dlr):
        self.plan_req.reduce_wide(self.scratch_partials, gout)

    def pack(self, route, gbuf):
        call("rs_shard_pack", ptr(self.plan_req.plan), ptr(route), self.F, self.B, self.d,
             ptr(self.plan_req.rowgrad_wide), None, ptr(gbuf), None, self.W, 1, stream())

    def merge(self, recv_idx, gbuf, partials):
        self.plan_own.reduce_wide(partials, gbuf.view(-1, self.d))

    def update(self, opt, hyper, scal, table, m, v, table1, m1, v1, rs=None, rs1=None):
        self.plan_own.update_wide(opt, hyper, scal, table, m, v, rs=rs)

    @property
    def n_own_partials(self):
        return self.plan_own.n_partials_wide


# -------------------------------------------------------------------------------------------------
# the exchange protocol
# -------------------------------------------------------------------------------------------------
class ShardedExchange:
    """Three all-to-alls per step and table set: distinct keys out, rows back, one gradient per distinct key out.
    Rows and gradients of the d-wide table and of the 1-dim (LR) table share one buffer (payload_width)."""

    def __init__(self, kernels, comm: Comm, spec: ShardSpec, B: int, d: int, with_emb: bool, with_lr: bool,
                 buckets: BucketLayout, device):
        self.k, self.comm, self.spec = kernels, comm, spec
        N, F = spec.N, len(spec.vocab)
        self.N, self.F, self.B, self.d, self.bk = N, F, B, d, buckets
        self.with_emb, self.with_lr = with_emb, with_lr
        dev = device
        S = buckets.S
        W = kernels.W
        self.W = W
        z = lambda shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=dev)
        self.send_idx = z((N, S + F), torch.int64)
        self.recv_idx = z((N, S + F), torch.int64)
        self.route = torch.full((F, B), -1, dtype=torch.int32, device=dev)
        self.overflow = z((1,), torch.int32)
        self.pay_s = z((N * S, W))
        self.pay_r = z((N * S + 1, W))          # position N*S stays zero: what padding entries and lost keys read
        self.g_s = z((N * S, W))
        self.g_r = z((N * S, W))

    # ---- forward, in the three pieces a trainer may interleave with other work -----------------------------
    def send_keys(self, idx, comm: Optional[Comm] = None):
        """Requester: sort + de-duplicate the batch, bucket its distinct keys by owner, exchange them (comm: another
        communicator than the exchange's own, for a key exchange that runs beside the step before)."""
        k, c = self.k, comm or self.comm
        k.sort_batch(idx)
        k.route(self.send_idx, self.route, self.overflow)
        c.all_to_all(self.recv_idx, self.send_idx)
        if hasattr(k, "counts"):
            k.counts(self.recv_idx)
        if hasattr(k, "merge_plan"):
            k.merge_plan(self.recv_idx)

    def serve_rows(self, table, table1, rs=None, rs1=None):
        """Owner: bring the requested rows up to date (exact Adam), look them up, send them back."""
        k, c = self.k, self.comm
        served = False
        if rs is not None or rs1 is not None:
            served = (getattr(k, "serves_in_catch_up", False) and (rs is not None) == (table is not None)
                      and (rs1 is not None) == (table1 is not None))
            if served:
                k.catch_up(self.recv_idx, rs, rs1, pay=self.pay_s)
            else:
                k.catch_up(self.recv_idx, rs, rs1)
        if not served:
            k.gather(self.recv_idx, table, table1, self.pay_s)
        c.all_to_all(self.pay_r[:self.N * self.bk.S], self.pay_s)

    def forward(self, idx, table, table1, emb_out, lr1_out, rs=None, rs1=None):
        """idx [F,B] global indices -> emb_out [B*F, d] rows, lr1_out [B*F] 1-dim values.  rs / rs1: ops.RowAdam of
        table / table1 when the owner replays skipped Adam steps of the requested rows before serving them."""
        self.send_keys(idx)
        self.serve_rows(table, table1, rs, rs1)
        self.k.expand(self.route, self.pay_r, emb_out, lr1_out)

    def backward(self, gout, dfm, S, emb, dlr, own_partials, colsum=None, splitk=None):
        """Per-entry gradient terms of the local batch -> summed row gradients on the owners."""
        self.reduce_local(gout, dfm, S, emb, dlr, colsum=colsum, splitk=splitk)
        self.send_grads(own_partials)

    def reduce_local(self, gout, dfm, S, emb, dlr, colsum=None, splitk=None):
        kw = {}
        if colsum is not None or splitk is not None:
            kw.update(colsum=colsum, splitk=splitk)
        if getattr(self.k, "packs_in_reduce", False):
            kw.update(route=self.route, gbuf=self.g_s)
        self.k.reduce_local(gout, dfm, S, emb, dlr, **kw)

    def send_grads(self, own_partials, before_merge=None, **merge_kw):
        """before_merge: called between the all-to-all and the owner's reduction (a collective of the caller's that
        the reduction's launches then build on)."""
        k, c = self.k, self.comm
        if not getattr(k, "packs_in_reduce", False):
            k.pack(self.route, self.g_s)
        c.all_to_all(self.g_r, self.g_s)
        if before_merge is not None:
            before_merge()
        k.merge(self.recv_idx, self.g_r, own_partials, **merge_kw)

    def check_overflow(self):
        """Host check (device sync): a bucket overflow drops rows; raise so the caller can re-plan."""
        if int(self.overflow.item()) != 0:
            self.overflow.zero_()
            raise RSError("row-shard exchange: a (rank, field) bucket overflowed its capacity; "
                          "raise bucket_slack")


def _settle_collectives():
    """The warm-up steps of a capture issue their collectives for real; the process group's watchdog polls the events of
    those every 100 ms until they are done, and must be through with them before the streams they ran on start
    capturing."""
    import time
    torch.cuda.synchronize()
    time.sleep(0.3)


def _capture_with_collectives(trainer, idx, labels, next_idx=None, entry=None, run_prologue=False):
    """FusedTrainer._capture for a step with RCCL collectives inside.  The warm-up steps (and the first replay) run for
    real -- collectives included, on every rank -- and are undone.  Other threads of the process group (its watchdog)
    keep working during capture, hence the thread-local capture mode; the watchdog is given time to retire the
    collectives issued so far before the capture begins (it polls their events, and must not find one that the capture
    has taken over)."""
    import time
    trainer._capture_error_mode = "thread_local"
    torch.cuda.synchronize()
    time.sleep(0.5)
    return FusedTrainer._capture(trainer, idx, labels, next_idx, entry, run_prologue)


# -------------------------------------------------------------------------------------------------
# the distributed fused step
# -------------------------------------------------------------------------------------------------
class DistributedTrainer(FusedTrainer):
    """FusedTrainer with row-sharded tables and a data-parallel dense tower (see the module docstring).

    step_tensors(..., next_idx=...) (the batch of the next call, see FusedTrainer.step_tensors): the key exchange of
    that batch -- its sort, the bucketing of its distinct keys, their all-to-all, the owners' list lengths -- runs on the
    second stream and communicator beside this step's GEMMs, and the next step starts with the owners serving rows.

    Every rank must build the same model from the same seed before constructing the trainer; the
    trainer keeps only this rank's shard of each table as training state.  ``sync_to_model()``
    all-gathers the shards back into the model's parameters (for evaluation / checkpoints).

    One step, after the dense backward:  the requester's reduction (one gradient per distinct key; the finishing sums
    of the dense gradients ride in its launch) -> on a second stream and a second communicator the all-reduce of the flat
    dense gradient and its norm partials, beside, on the main stream, pack -> all-to-all of the gradients -> owner-side
    merge + reduction -> all-reduce of the owners' ||g||^2 partials -> join -> clip coefficient -> ONE launch for the
    shard update and the dense update.  use_graph: the whole step, collectives included, is one HIP graph.
    """

    def __init__(self, model, mode: str = "sparse", use_graph: bool = False, group=None, stage_cpu: bool = False,
                 bucket_slack: float = 1.5, overlap_comm: Optional[bool] = None, **kw):
        if mode != "sparse":
            raise RSError("DistributedTrainer: row-sharded tables use the touched-rows (sparse) update")
        if use_graph and stage_cpu:
            raise RSError("DistributedTrainer: collectives staged through host memory (gloo) cannot be captured in a HIP graph")
        self.comm = Comm(group, stage_cpu=stage_cpu)
        self.world, self.rank = self.comm.world, self.comm.rank
        self.bucket_slack = bucket_slack
        super().__init__(model, mode="sparse", use_graph=use_graph, **kw)
        if self.overlap:
            raise RSError("DistributedTrainer: side streams for the weight gradients (overlap) do not go with the exchange")
        self._sort_can_fork = False
        self._update_shares_launch = True
        self._reduce_carries_finishes = True
        # the all-reduce of the dense gradient on a stream and a communicator of its own, beside the gradient exchange
        # (the same streams carry the key exchange of an announced next batch, see step_tensors)
        self.overlap_comm = True if overlap_comm is None else bool(overlap_comm)
        self.comm_dense = self.comm.second() if self.overlap_comm else self.comm
        self.side_comm = torch.cuda.Stream(device=self.device) if self.overlap_comm else None
        self.ws_comm = ops.Workspace(self.device)
        # the key exchange of an announced next batch: a stream of its own, forked once and joined once per step
        # (and a communicator of its own: one communicator per stream)
        self.side_keys = torch.cuda.Stream(device=self.device) if self.overlap_comm else None
        self.comm_keys = self.comm.second() if self.overlap_comm else self.comm
        self.ws_keys = ops.Workspace(self.device)
        self._keys_in_flight = False
        self._key_graphs = {}
        self._keys_event = None
        # a bucket that overflows serves zeros for the keys it lost: look at the flags often (they are read without a
        # host synchronisation, on a stream of their own, and reported one to three polls after the device raised them:
        # within 15 steps of the step that overflowed)
        self.check_every = 5
        vocab = self.layout.vocab_sizes
        self.spec = ShardSpec(vocab, self.world, self.rank)
        # this rank's shards (training state) and their optimizer moments
        adam = self.opt in ("Adam", "AdamW")
        need_m = adam or self.opt == "Adagrad"
        zl = torch.zeros_like
        self.shard = self.spec.shard_table(self.tab.stacked) if self.has_emb else None
        self.shard1 = self.spec.shard_table(self.tab1.stacked).view(-1) if self.has_lr else None
        self.shard_m = zl(self.shard) if (self.has_emb and need_m) else None
        self.shard_v = zl(self.shard) if (self.has_emb and adam) else None
        self.shard1_m = zl(self.shard1) if (self.has_lr and need_m) else None
        self.shard1_v = zl(self.shard1) if (self.has_lr and adam) else None
        for t in self.tables:            # the replicated full tables carry no optimizer state here
            t["m"] = t["v"] = None
        # exact Adam on touched rows: the per-row step stamps live with the shards, on their owner
        self.rs_shard = self.rs_shard1 = None
        if self.exact:
            if self.has_emb:
                self.rs_shard = ops.RowAdam(self.shard, self.shard_m, self.shard_v, self.opt, self.hyper, self.step_count)
            if self.has_lr:
                self.rs_shard1 = ops.RowAdam(self.shard1, self.shard1_m, self.shard1_v, self.opt, self.hyper, self.step_count,
                                             bctab=self.rs_shard.bctab if self.rs_shard is not None else None)
        self._ex = {}
        # the dense tower must start identical on every rank
        self.comm.broadcast(self.flat_w, 0)

    _ahead_kind = "keys"

    def _step_impl(self, idx, labels, next_idx=None):
        # BatchNorm layers take their batch statistics over the union of the ranks' batches (two small collectives per
        # layer and step on the main communicator), so the replicas' running statistics stay identical
        with ops.use_bn_sync(self.comm):
            return super()._step_impl(idx, labels, next_idx)

    def _ahead_supported(self, B: int) -> bool:
        return self.overlap_comm             # the key exchange of the next batch needs the second stream and communicator

    def _run_prologue(self, idx, entry):
        self.exchange(entry["B"], entry["slot"]).send_keys(idx)

    def exchange(self, B: int, slot: int = 0) -> ShardedExchange:
        """Buffers and plans of the exchange for batches of B samples; two slots, so that the keys of the next batch
        can be exchanged while this one's are still needed by its backward pass."""
        key = (B, slot)
        if key not in self._ex:
            bk = self.spec.buckets(B, slack=self.bucket_slack, device=self.device)
            k = HipShardKernels(self.spec, B, self.d, self.has_emb, self.has_lr, bk, self.device)
            ex = ShardedExchange(k, self.comm, self.spec, B, self.d, self.has_emb, self.has_lr, bk, self.device)
            # the partial sums of the global gradient norm, in one buffer: ||g||^2 of the rows this rank owns (summed
            # over ranks by an all-reduce of its own: ranks own disjoint rows), then those of the flat dense gradient
            ex.norm_buf = torch.zeros(k.n_own_partials + N_DENSE_PARTIALS, dtype=torch.float32, device=self.device)
            ex.own_partials = ex.norm_buf[:k.n_own_partials]
            ex.dense_partials = ex.norm_buf[k.n_own_partials:]
            self._ex[key] = ex
        return self._ex[key]

    # ---- hooks -----------------------------------------------------------------------------------
    def _grad_scale(self) -> float:
        return 1.0 / self.world          # mean over the N*B samples of the global batch

    def _attach_row_state(self):
        """The model's full tables are not trained here (the shards are): they carry no step stamps."""

    def _embed_catch_up(self, idx, mark_ahead: int):
        """The owners replay skipped steps inside the exchange, once they know which rows are requested."""

    def flush(self):
        for rs in (self.rs_shard, self.rs_shard1):
            if rs is not None:
                rs.flush()

    def _embed_forward(self, idx):
        """K1 through the exchange: distinct keys out, rows back, then the fused lookup in the received rows."""
        c, B = self._step_ctx, idx.shape[1]
        ex = self._cur_ex = self.exchange(B, c["slot"])
        if not c["done_ahead"]:
            ex.send_keys(idx)
        if c["next_idx"] is not None:
            # the key exchange of the next batch, beside this step: second stream, second communicator, the other slot
            nxt = self.exchange(B, 1 - c["slot"])
            self.side_keys.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side_keys), ops.use_workspace(self.ws_keys):
                nxt.send_keys(c["next_idx"], comm=self.comm_keys)
            self._keys_in_flight = True
        ex.serve_rows(self.shard, self.shard1, rs=self.rs_shard, rs1=self.rs_shard1)
        return ex.k.lookup(ex.pay_r, self.lrm.bias.data if self.has_lr else None, want_emb=self.has_emb,
                           want_sum=self.has_fm, want_lr=self.has_lr, want_fm=self.has_fm)

    def _embed_backward(self, plan, idx, gout, dz, ctx, sorted_already, dense_partials=None, colsum=None, splitk=None):
        B = idx.shape[1]
        ex = self._cur_ex
        ex.reduce_local(gout, dz if self.has_fm else None, ctx["S"],
                        ctx["emb"].view(B * self.F, self.d) if self.has_fm else None, dz if self.has_lr else None,
                        colsum=colsum, splitk=splitk)
        # the flat dense gradient is final from here on (its finishing sums rode in the launch above)
        if self.overlap_comm:
            main = torch.cuda.current_stream()
            self.side_comm.wait_stream(main)
            with torch.cuda.stream(self.side_comm), ops.use_workspace(self.ws_comm):
                self._dense_grad_norm(ex)
            ex.send_grads(ex.own_partials)
            self.comm.all_reduce(ex.own_partials)
            main.wait_stream(self.side_comm)
            if self._keys_in_flight:
                main.wait_stream(self.side_keys)
                self._keys_in_flight = False
        else:
            # one stream: the dense gradient is summed over ranks in front of the owner's reduction, whose second
            # launch then also leaves its norm partials
            ex.send_grads(ex.own_partials, before_merge=lambda: self._dense_grad_norm(ex, sumsq=False),
                          dense=self.flat_g, dense_partials=ex.dense_partials)
            self.comm.all_reduce(ex.own_partials)

    def _dense_grad_norm(self, ex, sumsq=True):
        """Sum the flat dense gradient over ranks; its L2 term; its sum-of-squares partials."""
        self.comm_dense.all_reduce(self.flat_g)
        if self.net_reg:
            ops.axpy(self.net_reg, self.flat_w, self.flat_g)
        if sumsq:
            ops.sumsq_partials(self.flat_g, ex.dense_partials)

    def _grad_norm_partials(self, plan, fuse, dense_p):
        return self._cur_ex.norm_buf

    def _embed_update(self, plan, with_dense=False, stamps=True, next_plan=None):
        # `plan` is the base class's per-batch-size plan; the update runs on the owner-side plan of the exchange
        ex = self._cur_ex
        ex.k.update(self.opt, self.hyper, self.scal, self.shard, self.shard_m, self.shard_v, self.shard1,
                    self.shard1_m, self.shard1_v,
                    dense=(self.flat_w, self.flat_g, self.flat_m, self.flat_v) if with_dense else None,
                    rs=self.rs_shard, rs1=self.rs_shard1)

    # ---- the whole step as one HIP graph, collectives inside --------------------------------------------
    def _capture(self, idx, labels, next_idx=None, entry=None, run_prologue=False):
        return _capture_with_collectives(self, idx, labels, next_idx, entry, run_prologue)

    def _before_capture(self):
        _settle_collectives()

    def step_tensors(self, idx, labels, resident: bool = False, next_idx=None):
        """FusedTrainer.step_tensors.  With use_graph, an announced next batch gets a HIP graph of its own for its key
        exchange, replayed on the key stream beside the step's graph (forking a captured step into the key stream early
        and joining it late makes hipStreamEndCapture crash on ROCm 7.0)."""
        if self._keys_event is not None:                 # the keys this step may start from have arrived
            torch.cuda.current_stream().wait_event(self._keys_event)
            self._keys_event = None
        B = idx.shape[1]
        if not (self.use_graph and resident and next_idx is not None and self._ahead_supported(B)
                and tuple(next_idx.shape) == tuple(idx.shape) and next_idx.dtype == torch.int64 and next_idx.is_contiguous()):
            return FusedTrainer.step_tensors(self, idx, labels, resident=resident, next_idx=next_idx)
        slot = self._ahead_slot_for(idx)
        cur = slot if slot is not None else 0
        self._keys_graph(next_idx, 1 - cur)
        loss = FusedTrainer.step_tensors(self, idx, labels, resident=True, next_idx=None)
        self._ahead = {"key": self._tensor_key(next_idx), "slot": 1 - cur, "B": B, "marked": False}
        return loss

    def _keys_graph(self, next_idx, slot):
        """send_keys of the announced batch into exchange slot `slot`, on the key stream, from a graph of its own."""
        import time
        ex = self.exchange(next_idx.shape[1], slot)
        side = self.side_keys
        side.wait_stream(torch.cuda.current_stream())    # the step that last used this slot's buffers has been issued
        key = (next_idx.data_ptr(), tuple(next_idx.shape), slot)
        with torch.cuda.stream(side), ops.use_workspace(self.ws_keys):
            g = self._key_graphs.get(key)
            if g is None:
                ex.send_keys(next_idx, comm=self.comm_keys)          # for real (and the kernels' first launch)
                torch.cuda.synchronize()
                time.sleep(0.5)                                       # see _capture_with_collectives
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    ex.send_keys(next_idx, comm=self.comm_keys)
                self._key_graphs[key] = g
            else:
                g.replay()
            self._keys_event = torch.cuda.Event()
            self._keys_event.record(side)

    def _mutable_state(self):
        ts = super()._mutable_state()
        for x in (self.shard, self.shard1, self.shard_m, self.shard_v, self.shard1_m, self.shard1_v):
            if x is not None:
                ts.append(x)
        for rs in (self.rs_shard, self.rs_shard1):
            if rs is not None:
                ts += rs.tensors()
        return ts

    # ---- utilities ---------------------------------------------------------------------------------
    def sync_to_model(self):
        """All-gather the shards into the model's (replicated) embedding parameters."""
        self.check()
        self.flush()
        if self.has_emb:
            self.spec.unshard_into(self.tab.stacked, self._gather_shards(self.shard))
        if self.has_lr:
            self.spec.unshard_into(self.tab1.stacked, [s.view(-1, 1) for s in self._gather_shards(self.shard1.view(-1, 1))])

    def _gather_shards(self, shard):
        # shards differ in length by at most F rows across ranks: pad to the longest
        n = torch.tensor([shard.shape[0]], dtype=torch.int64, device=shard.device)
        ns = [int(x.item()) for x in self.comm.all_gather(n)]
        mx = max(ns)
        pad = torch.zeros((mx,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
        pad[:shard.shape[0]] = shard
        return [g[:k] for g, k in zip(self.comm.all_gather(pad), ns)]

    def _flag_tensors(self):
        out = []
        for ex in self._ex.values():
            out += [(ex.overflow, "bucket overflow"), (ex.k.layout_own.err_flag, "index (owner)"),
                    (ex.k.layout_req.err_flag, "index (requester)")]
        return out           # (the merge's capacity marker, which the bucket sizes rule out, is left to check())

    def check(self):
        """Bucket overflow of the exchange (rows would be served as zeros and their gradients dropped) and index range:
        raise.  Synchronous; every check_every steps the flags are polled without synchronisation (_poll_flags), and this
        runs when a poll finds one raised, and before sync_to_model()."""
        self._unchecked = 0
        for ex in self._ex.values():
            ex.check_overflow()
            ex.k.layout_own.check_indices()
            ex.k.layout_req.check_indices()
            ex.k.plan_own.check()


# -------------------------------------------------------------------------------------------------
# the replicated-tables variant
# -------------------------------------------------------------------------------------------------
class HipReplicaKernels:
    """Device side of the replicated-tables merge on librs_hip.so (tests/test_dist_cpu.py substitutes a torch-CPU
    stand-in with the same methods)."""

    def __init__(self, layout: FieldLayout, N: int, B: int, d: int, with_emb: bool, with_lr: bool, device):
        self.N, self.B, self.d, self.F = N, B, d, layout.num_fields
        self.with_emb, self.with_lr = with_emb, with_lr
        self.gplan = SparsePlan(layout, cap=N * B, d=d, with_lr=with_lr)
        sf = torch.arange(N * self.F, dtype=torch.int32, device=device)
        self.sub_off = sf * B
        self.pay_base = sf * B
        self.row_off = layout.row_offset_t.view(self.F, 1)

    @property
    def n_partials(self):
        return self.gplan.n_partials

    def keys_of(self, seg_rows, out):
        """stacked-table rows [F, B] (int32) -> keys inside their field (int64)."""
        torch.sub(seg_rows.long(), self.row_off, out=out)

    def merge(self, keys_all, cnt_all, g_all, g1_all, partials):
        """N ascending key lists per field + their gradients -> one gradient per distinct row (gplan.rowgrad /
        rowgrad1, rows in gplan.seg_rows()), ||g||^2 partials."""
        N, B = self.N, self.B
        self.gplan.merge_lists(keys_all.view(-1), self.sub_off, cnt_all.view(-1), self.pay_base, nsrc=N, max_entries=N * B)
        self.gplan.reduce(partials, gout=g_all.view(-1, self.d) if g_all is not None else None,
                          g1src=g1_all.view(-1) if g1_all is not None else None, Fdiv=1)


class ReplicaExchange:
    """All-gather of every rank's distinct (row, gradient) pairs -- per field an ascending key list of at most B
    entries with its summed gradients, rectangular buffers -- and their merge in rank order."""

    def __init__(self, kernels, comm: Comm, F: int, B: int, d: int, with_emb: bool, with_lr: bool, device):
        self.k, self.comm = kernels, comm
        N = comm.world
        z = lambda shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=device)
        self.keys = z((F, B), torch.int64)
        self.keys_all = z((N, F, B), torch.int64)
        self.cnt = z((F,), torch.int32)
        self.cnt_all = z((N, F), torch.int32)
        self.g_all = z((N, F, B, d)) if with_emb else None
        self.g1_all = z((N, F, B)) if with_lr else None
        self.partials = z((kernels.n_partials,))

    def merge(self, seg_rows, n_seg, rowgrad, rowgrad1):
        """seg_rows [F, B] int32 stacked rows (ascending per field, n_seg[f] valid), rowgrad [F, B, d] / rowgrad1
        [F, B]: this rank's distinct rows and gradients.  Leaves the merged result in the kernels object."""
        c = self.comm
        self.k.keys_of(seg_rows, self.keys)
        self.cnt.copy_(n_seg)
        c.all_gather_into(self.keys_all, self.keys)
        c.all_gather_into(self.cnt_all, self.cnt)
        if self.g_all is not None:
            c.all_gather_into(self.g_all, rowgrad)
        if self.g1_all is not None:
            c.all_gather_into(self.g1_all, rowgrad1)
        self.k.merge(self.keys_all, self.cnt_all, self.g_all, self.g1_all, self.partials)



class ReplicatedTablesTrainer(FusedTrainer):
    """Data-parallel step with every table replicated on every rank (for tables that fit one GPU).

    Forward and backward are the single-GPU step on the rank's own batch.  Then, instead of an all-reduce of dense
    table gradients, the (row, gradient) pairs of the DISTINCT rows each rank touched are all-gathered -- per field
    an ascending key list and its summed gradients, rectangular buffers of the batch size -- merged per field
    (`SparsePlan.merge_lists`: the N lists are sorted already) and summed in rank order, and every rank applies the
    same touched-rows update to its replica, so the replicas stay bitwise identical without ever being exchanged.
    The global gradient norm needs no collective of its own: every rank holds every table gradient after the merge.
    Collectives per step: one all-reduce of the flat dense gradient, one all-gather each of keys, counts, row
    gradients (and 1-dim gradients).  Limit: world * batch entries per field <= rs_merge_max_entries()."""

    def __init__(self, model, use_graph: bool = False, group=None, stage_cpu: bool = False, **kw):
        if use_graph and stage_cpu:
            raise RSError("ReplicatedTablesTrainer: collectives staged through host memory (gloo) cannot be captured in a HIP graph")
        self.comm = Comm(group, stage_cpu=stage_cpu)
        self.world, self.rank = self.comm.world, self.comm.rank
        kw.pop("mode", None)
        super().__init__(model, mode="sparse", use_graph=use_graph, **kw)
        self._sort_can_fork = False
        self._update_shares_launch = True
        self._reduce_carries_finishes = True
        self._rp = {}
        self.comm.broadcast(self.flat_w, 0)
        for t in self.tables:
            self.comm.broadcast(t["w"], 0)

    def _step_impl(self, idx, labels, next_idx=None):
        with ops.use_bn_sync(self.comm):          # BatchNorm statistics over the union of the ranks' batches
            return super()._step_impl(idx, labels, next_idx)

    def _replica(self, B: int) -> "ReplicaExchange":
        if B not in self._rp:
            k = HipReplicaKernels(self.layout, self.world, B, self.d, self.has_emb, self.has_lr, self.device)
            self._rp[B] = ReplicaExchange(k, self.comm, self.F, B, self.d, self.has_emb, self.has_lr, self.device)
        return self._rp[B]

    def _grad_scale(self) -> float:
        return 1.0 / self.world

    def _embed_backward(self, plan, idx, gout, dz, ctx, sorted_already, dense_partials=None, colsum=None, splitk=None):
        B = idx.shape[1]
        if plan.cap != B:
            raise RSError("replicated tables: the plan capacity must equal the batch size")
        FusedTrainer._embed_backward(self, plan, idx, gout, dz, ctx, sorted_already, colsum=colsum,
                                     splitk=splitk)                                    # one gradient per distinct row
        rx = self._replica(B)
        rx.merge(plan.seg_rows(), plan.counts()[:, 2], plan.rowgrad if self.has_emb else None,
                 plan.rowgrad1 if self.has_lr else None)
        if self.exact:
            # the update steps the union of every rank's rows: the ones only other ranks read are brought up to date here
            ops.adam_catchup_lists(rx.keys_all.view(-1), rx.k.sub_off, rx.cnt_all.view(-1), B, self.layout, *self._row_states())
        P = plan.partials
        P[:plan.n_partials].zero_()
        ops.colsum(rx.partials.view(-1, 1), out=P[0:1])

    def _reduce_dense_grads(self):
        self.comm.all_reduce(self.flat_g)

    def _embed_update(self, plan, with_dense=False):
        FusedTrainer._embed_update(self, self._rp[plan.cap].k.gplan, with_dense=with_dense)

    def _capture(self, idx, labels, next_idx=None, entry=None, run_prologue=False):
        return _capture_with_collectives(self, idx, labels)

    def _before_capture(self):
        _settle_collectives()

    def sync_to_model(self):
        """Nothing to gather: the model's tables are this rank's replica (rows left behind are brought up to date)."""
        self.flush()

    def check(self):
        self._unchecked = 0
        self.layout.check_indices()
        for rx in self._rp.values():
            rx.k.gplan.check()


# -------------------------------------------------------------------------------------------------
# the distributed generic step (any model kind)
# -------------------------------------------------------------------------------------------------
class _ShardedLookups:
    """TableCapture provider: rows and LR terms come from the row-sharded tables through the exchange."""

    def __init__(self, trainer):
        self.t = trainer

    def emb(self, idx, holder):
        st, ex = self.t._sharded(holder, "emb", idx.shape[1])
        F, B = idx.shape
        out = torch.empty((B, F, holder.dim), dtype=torch.float32, device=idx.device)
        ex.forward(idx, st["shard"], None, out.view(B * F, holder.dim), None, rs=st["rs"])
        return out

    def ffm_rows(self, idx, holder):
        st, ex = self.t._sharded(holder, "ffm", idx.shape[1])
        F, B = idx.shape
        out = torch.empty((B, F, holder.dim), dtype=torch.float32, device=idx.device)
        ex.forward(idx, st["shard"], None, out.view(B * F, holder.dim), None, rs=st["rs"])
        return out

    def lr_terms(self, idx, holder):
        st, ex = self.t._sharded(holder, "lr", idx.shape[1])
        F, B = idx.shape
        out = torch.empty((B, F), dtype=torch.float32, device=idx.device)
        ex.forward(idx, None, st["shard"], None, out.view(-1), rs1=st["rs"])
        return out


class DistributedSparseTrainer(SparseTrainer):
    """SparseTrainer over row-sharded tables and a data-parallel dense part: the multi-GPU step for the model kinds
    the fused engine does not schedule (PNN, NFM, AFM, DCN, DeepCrossing, ...).  Same protocol as DistributedTrainer:
    per looked-up table three all-to-alls (distinct keys out, rows back, one gradient per distinct key out), one
    all-reduce of the flat dense gradient with the tables' ||g||^2 riding behind it, mean loss over the N*B samples.
    Every rank must build the same model from the same seed.  FFM's field-aware rows (F*d floats each) travel the same
    way through the wide-row kernels (HipShardKernelsWide); the FFM term is then evaluated on the gathered rows."""

    def __init__(self, model, group=None, stage_cpu: bool = False, bucket_slack: float = 1.5, **kw):
        if kw.pop("use_graph", False):
            raise RSError("DistributedSparseTrainer: the step is issued eagerly (collectives between the launches)")
        self.comm = Comm(group, stage_cpu=stage_cpu)
        self.world, self.rank = self.comm.world, self.comm.rank
        self.bucket_slack = bucket_slack
        super().__init__(model, **kw)
        self._shards = {}
        self._provider = _ShardedLookups(self)
        self.comm.broadcast(self.flat_w, 0)          # the dense part must start identical on every rank

    def _sharded(self, holder, kind, B):
        st = self._shards.get(id(holder))
        if st is None:
            spec = ShardSpec(holder.layout.vocab_sizes, self.world, self.rank)
            shard = spec.shard_table(holder.stacked)
            if kind == "lr":
                shard = shard.view(-1)
            st = {"holder": holder, "kind": kind, "spec": spec, "shard": shard,
                  "m": torch.zeros_like(shard) if self.need_m else None,
                  "v": torch.zeros_like(shard) if self.need_v else None, "ex": {}, "rs": None}
            if self.exact:      # per-row step stamps of the shard, on its owner
                st["rs"] = ops.RowAdam(shard, st["m"], st["v"], self.opt, self.hyper, self.step_count, bctab=self._bctab)
                self._bctab = st["rs"].bctab
            self._shards[id(holder)] = st
        ex = st["ex"].get(B)
        if ex is None:
            spec = st["spec"]
            bk = spec.buckets(B, slack=self.bucket_slack, device=self.device)
            emb = kind != "lr"
            if kind == "ffm":
                k = HipShardKernelsWide(spec, B, holder.dim, bk, self.device)
            else:
                k = HipShardKernels(spec, B, holder.dim if emb else 1, emb, not emb, bk, self.device)
            ex = ShardedExchange(k, self.comm, spec, B, holder.dim if emb else 1, emb, not emb, bk, self.device)
            ex.own_partials = torch.zeros(k.n_own_partials, dtype=torch.float32, device=self.device)
            st["ex"][B] = ex
        return st, ex

    def _step_impl(self, inputs, y) -> torch.Tensor:
        with ops.use_bn_sync(self.comm):          # BatchNorm statistics over the union of the ranks' batches
            cap, loss = self._forward_backward(inputs, y, provider=self._provider, grad_scale=1.0 / self.world)
        jobs, seen = [], set()
        for rec in cap.records:
            g = rec["out"].grad
            if g is None:
                continue
            if id(rec["holder"]) in seen:
                raise RSError("DistributedSparseTrainer: a table was looked up twice in one forward pass")
            seen.add(id(rec["holder"]))
            B = rec["idx"].shape[1]
            st, ex = self._sharded(rec["holder"], rec["kind"], B)
            if rec["kind"] == "emb":
                ex.backward(g.contiguous().view(B * ex.F, ex.d), None, None, None, None, ex.own_partials)
            elif rec["kind"] == "ffm":
                d = rec["holder"].dim // ex.F
                dG = ops.ffm_rows_bwd(rec["rows"], g.contiguous().view(B), d)
                ex.backward(dG.view(B * ex.F, ex.d), None, None, None, None, ex.own_partials)
            else:
                ex.backward(None, None, None, None, g.contiguous().view(B), ex.own_partials)
            jobs.append((st, ex))
        # ||g||^2 of the owned rows of every table (ranks own disjoint rows) rides behind the dense gradients
        if jobs:
            allp = torch.cat([ex.own_partials for _, ex in jobs]) if len(jobs) > 1 else jobs[0][1].own_partials
            ops.colsum(allp.view(-1, 1), out=self.flat_g_ext[-4:-3])
        self.comm.all_reduce(self.flat_g_ext)
        if self._partials is None or self._partials.numel() != N_DENSE_PARTIALS + 1:
            self._partials = torch.zeros(N_DENSE_PARTIALS + 1, dtype=torch.float32, device=self.device)
        P = self._partials
        P[0:1].copy_(self.flat_g_ext[-4:-3])
        ops.sumsq_partials(self.flat_g, P[1:])
        ops.clip_and_prepare(P, self.max_norm, self.scal, self.step_count, self.hyper)
        ops.dense_update(self.flat_w, self.flat_g, self.flat_m, self.flat_v, self.opt, self.hyper, self.scal)
        for st, ex in jobs:
            if st["kind"] != "lr":
                ex.k.update(self.opt, self.hyper, self.scal, st["shard"], st["m"], st["v"], None, None, None, rs=st["rs"])
            else:
                ex.k.update(self.opt, self.hyper, self.scal, None, None, None, st["shard"], st["m"], st["v"], rs1=st["rs"])
        self.last = {"records": cap.records, "direct_grads": len(cap._sunk)}
        return loss.detach()

    def flush(self):
        for st in self._shards.values():
            if st["rs"] is not None:
                st["rs"].flush()

    def sync_to_model(self):
        """All-gather the shards into the model's (replicated) table parameters."""
        self.check()
        self.flush()
        for st in self._shards.values():
            shard = st["shard"].view(-1, 1) if st["kind"] == "lr" else st["shard"]
            n = torch.tensor([shard.shape[0]], dtype=torch.int64, device=shard.device)
            ns = [int(x.item()) for x in self.comm.all_gather(n)]
            pad = torch.zeros((max(ns),) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
            pad[:shard.shape[0]] = shard
            st["spec"].unshard_into(st["holder"].stacked, [g[:k] for g, k in zip(self.comm.all_gather(pad), ns)])

    def check(self):
        self._unchecked = 0
        for st in self._shards.values():
            for ex in st["ex"].values():
                ex.check_overflow()
                ex.k.layout_own.check_indices()
                ex.k.layout_req.check_indices()
                ex.k.plan_own.check()
