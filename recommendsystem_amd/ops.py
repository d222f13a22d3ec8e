"""Tensor-level wrappers over the C-ABI (no autograd here; see functional.py).

Every function takes CUDA (ROCm) tensors, validates shapes/dtypes on the host so the kernels
never see an operand they were not written for, allocates outputs with torch, and launches
on torch's current stream.
"""
from __future__ import annotations

import os
from typing import List, Optional

import contextlib

import torch

from . import _lib
from ._lib import call, ptr, stream, require_cuda, RSError


def _chk(t: Optional[torch.Tensor], dtype, name: str, shape=None):
    if t is None:
        return
    if t.dtype != dtype:
        raise RSError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RSError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RSError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    require_cuda(t)


def _chk_rows(t: Optional[torch.Tensor], name: str, shape=None):
    """A float32 matrix whose rows are contiguous but may lie further apart than their length (a block of columns of
    a wider matrix): what the GEMM's leading-dimension arguments describe.  Returns the row stride in elements."""
    if t is None:
        return 0
    if t.dtype != torch.float32:
        raise RSError(f"{name}: expected dtype torch.float32, got {t.dtype}")
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1) or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        raise RSError(f"{name}: must be a 2-D tensor with contiguous rows")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RSError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    require_cuda(t)
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


class FieldLayout:
    """Where each field's rows sit in the stacked table: device copies of row_offset/vocab."""

    def __init__(self, vocab_sizes, device):
        self.vocab_sizes = [int(v) for v in vocab_sizes]
        off, tot = [], 0
        for v in self.vocab_sizes:
            off.append(tot)
            tot += v
        self.row_offsets = off
        self.total_rows = tot
        self.num_fields = len(off)
        self.device = torch.device(device)
        self.row_offset_t = torch.tensor(off, dtype=torch.int64, device=device)
        self.vocab_t = torch.tensor(self.vocab_sizes, dtype=torch.int64, device=device)
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=device)

    def check_indices(self):
        """Raise IndexError if any kernel saw an out-of-range index since the last check
        (device sync; the kernels clamp such indices to the padding row so they never fault)."""
        if int(self.err_flag.item()) != 0:
            self.err_flag.zero_()
            raise IndexError("index out of range in embedding lookup")


class FlagWatch:
    """Device-side error flags read without stopping the step stream: poll() looks at the copy it requested the time
    before (usually long since arrived in pinned host memory) and requests a new one.  A flag that is raised is
    therefore reported one to three polls after the device raised it; check() of the owner remains the synchronous
    form."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._host = None
        self._event = None
        self._stream = None
        self._pending = 0

    def poll(self, flags):
        """flags: list of (1-element int32 device tensor, nonzero -> name); returns the names found raised in the
        PREVIOUS snapshot (empty list if there was none yet), and requests a snapshot of the current values."""
        raised = []
        if self._event is not None:
            # The host may have issued many steps more than the device has run: a snapshot that has not arrived yet is
            # left pending (nothing new is requested) rather than waited for; the second poll in a row that finds it
            # pending does wait, which bounds how far the report can trail the steps the host has issued.
            if not self._event.query():
                self._pending += 1
                if self._pending < 2:
                    return raised
                self._event.synchronize()
            self._pending = 0
            vals = self._host.tolist()
            raised = [name for (t, name), v in zip(self._names, vals) if v != 0]
        n = len(flags)
        if n == 0:
            self._event = None
            return raised
        if self._host is None or self._host.numel() != n:
            self._host = torch.zeros(n, dtype=torch.int32).pin_memory()
        # on a stream of its own and without waiting for the step stream: flags are only ever raised by the kernels, so
        # a snapshot that misses one being raised right now shows it the next time -- and the step stream never waits
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._stream):
            snap = torch.cat([t.view(1) for t, _ in flags])
            self._host.copy_(snap, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        self._names = list(flags)
        return raised


def gather_fwd(idx: torch.Tensor, layout: FieldLayout, table: torch.Tensor) -> torch.Tensor:
    """out[b, f, :] = table[row_offset[f] + idx[f, b], :]; idx is [F, B] int64."""
    Fn, B = idx.shape
    _chk(idx, torch.int64, "idx", (layout.num_fields, B))
    d = table.shape[1]
    _chk(table, torch.float32, "table", (layout.total_rows, d))
    out = torch.empty((B, Fn, d), dtype=torch.float32, device=idx.device)
    call("rs_gather_fwd", ptr(idx), ptr(layout.row_offset_t), ptr(layout.vocab_t), ptr(table), ptr(out),
         B, Fn, d, ptr(layout.err_flag), stream())
    return out


def gather_lr_fm_fwd(idx: torch.Tensor, layout: FieldLayout, table: Optional[torch.Tensor],
                     table1: Optional[torch.Tensor], bias: Optional[torch.Tensor],
                     want_emb=True, want_sum=False, want_bi=False, want_lr=False, want_fm=False,
                     scale: Optional[torch.Tensor] = None):
    """Fused lookup; returns dict with the requested outputs (emb, sum, bi, lr, fm).
    scale: one float on the device every looked-up table value is multiplied by (TableShrink.gscale)."""
    Fn, B = idx.shape
    _chk(idx, torch.int64, "idx", (layout.num_fields, B))
    d = 1
    if table is not None:
        d = table.shape[1]
        _chk(table, torch.float32, "table", (layout.total_rows, d))
    if table1 is not None:
        if table1.dim() == 2 and table1.shape[1] == 1:
            table1 = table1.view(-1)
        _chk(table1, torch.float32, "table1", (layout.total_rows,))
    _chk(bias, torch.float32, "bias", (1,))
    dev = idx.device
    out = {}
    if want_emb:
        out["emb"] = torch.empty((B, Fn, d), dtype=torch.float32, device=dev)
    if want_sum:
        out["sum"] = torch.empty((B, d), dtype=torch.float32, device=dev)
    if want_bi:
        out["bi"] = torch.empty((B, d), dtype=torch.float32, device=dev)
    if want_lr:
        out["lr"] = torch.empty((B,), dtype=torch.float32, device=dev)
    if want_fm:
        out["fm"] = torch.empty((B,), dtype=torch.float32, device=dev)
    if scale is not None:
        _chk(scale, torch.float32, "scale", (1,))
        call("rs_gather_lr_fm_fwd_scaled", ptr(idx), ptr(layout.row_offset_t), ptr(layout.vocab_t), ptr(table), ptr(table1),
             ptr(bias), ptr(out.get("emb")), ptr(out.get("sum")), ptr(out.get("bi")), ptr(out.get("lr")),
             ptr(out.get("fm")), B, Fn, d, ptr(scale), ptr(layout.err_flag), stream())
        return out
    call("rs_gather_lr_fm_fwd", ptr(idx), ptr(layout.row_offset_t), ptr(layout.vocab_t), ptr(table), ptr(table1),
         ptr(bias), ptr(out.get("emb")), ptr(out.get("sum")), ptr(out.get("bi")), ptr(out.get("lr")),
         ptr(out.get("fm")), B, Fn, d, ptr(layout.err_flag), stream())
    return out


# ---------------------------------------------------------------------------------------------
# workspaces: named, cached device buffers (no allocation on the steady-state path)
# ---------------------------------------------------------------------------------------------
class Workspace:
    def __init__(self, device):
        self.device = torch.device(device)
        self._bufs = {}
        self._retired = []

    def get(self, name: str, numel: int, dtype=torch.float32, zero: bool = False) -> torch.Tensor:
        """zero=True: the buffer starts zeroed and its users leave it zeroed (arrival counters).  A buffer that has
        to grow is replaced, and the old one is kept alive: a captured HIP graph may still hold its address."""
        b = self._bufs.get(name)
        if b is None or b.numel() < numel or b.dtype != dtype:
            if b is not None:
                self._retired.append(b)
            mk = torch.zeros if zero else torch.empty
            b = mk(max(int(numel), 1), dtype=dtype, device=self.device)
            self._bufs[name] = b
        return b


_default_ws = {}
_ws_override = []


class use_workspace:
    """Context manager: route every op's scratch requests to `ws` (the engine keeps its scratch
    separate so that buffers captured in a HIP graph are never handed to eager callers)."""

    def __init__(self, ws: Workspace):
        self.ws = ws

    def __enter__(self):
        _ws_override.append(self.ws)
        return self.ws

    def __exit__(self, *exc):
        _ws_override.pop()
        return False


def default_workspace(device) -> Workspace:
    if _ws_override:
        return _ws_override[-1]
    device = torch.device(device)
    key = (device.type, device.index)
    if key not in _default_ws:
        _default_ws[key] = Workspace(device)
    return _default_ws[key]


ACT_CODES = {None: 0, "none": 0, "relu": 1, "tanh": 2, "leakyrelu": 3, "prelu": 4}
OPT_CODES = {"SGD": 0, "Adam": 1, "AdamW": 2, "Adagrad": 3}


# ---------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------
_gemm_plans = {}       # (M, N, K) -> (tile, splitk) of rs_gemm_plan


def gemm(a: torch.Tensor, b: torch.Tensor, *, a_mc=False, b_nc=False, bias=None, act=0, mulsrc=None,
         mul_act=0, out=None, tile=-1, splitk=0, ws: Optional[Workspace] = None, colsum_part=None,
         defer_reduce=None, accumulate=False, dropout=None) -> torch.Tensor:
    """C = epi(op(A) op(B)) in fp32 on the matrix cores.

    a_mc=False: a is [M, K];  a_mc=True: a is [K, M] (used as its transpose).
    b_nc=False: b is [N, K] (nn.Linear weight);  b_nc=True: b is [K, N].
    a, b, out and mulsrc may be blocks of columns of wider matrices (contiguous rows, any row stride).
    colsum_part: a ColsumParts collector; the GEMM leaves per-wave-row column sums of C in a scratch buffer and
    registers (buffer, rows, N, destination) with it -- ColsumParts.finish() adds them up in one launch for
    every registered GEMM (the bias gradients of the dZ's the backward chain produces).
    defer_reduce: a SplitKParts collector; a product whose reduction is split leaves its slabs there and `out` is
    only valid after the collector's finish() (one launch for all of them).
    accumulate: C = act(C + op(A) op(B) + bias) -- one term of a sum over blocks of the reduction dimension.
    dropout: (p, seed, salt, call_no, call_dev) as for dropout_apply -- the mask is multiplied into the result after bias
    and activation and before the mulsrc factor.
    """
    lda = _chk_rows(a, "a")
    ldb = _chk_rows(b, "b")
    M, K = (a.shape[1], a.shape[0]) if a_mc else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[1], b.shape[0]) if b_nc else (b.shape[0], b.shape[1])
    if K != Kb:
        raise RSError(f"gemm: inner dimensions differ ({K} vs {Kb})")
    _chk(bias, torch.float32, "bias", (N,))
    ldm = _chk_rows(mulsrc, "mulsrc", (M, N))
    if out is None:
        if accumulate:
            raise RSError("gemm: accumulate needs the running sum in `out`")
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ldc = _chk_rows(out, "out", (M, N))
    if M == 0 or N == 0:
        return out
    h = _lib.lib()
    if splitk <= 0 or tile < 0:
        plan = _gemm_plans.get((M, N, K))
        if plan is None:
            import ctypes as C
            sk = C.c_int(1)
            t = h.rs_gemm_plan(M, N, K, 256, C.byref(sk))
            plan = _gemm_plans[(M, N, K)] = (t, sk.value)
        if tile < 0:
            tile = plan[0]
        if splitk <= 0:
            splitk = plan[1]
    ws = ws or default_workspace(a.device)
    if (defer_reduce is not None and splitk > 1 and splitk * 32 <= K and bias is None and act == 0 and mulsrc is None
            and colsum_part is None and out.is_contiguous() and not accumulate):
        # slabs only; `out` is written by defer_reduce.finish()
        wbytes = h.rs_gemm_workspace_bytes(M, N, splitk)
        wbuf = defer_reduce.buffer(wbytes // 4)
        call("rs_gemm_f32_slabs", int(a_mc), int(b_nc), M, N, K, ptr(a), lda, ptr(b), ldb, int(tile),
             int(splitk), ptr(wbuf), wbytes, stream())
        defer_reduce.add(wbuf, out, M * N, splitk)
        return out
    wbuf, wbytes = None, 0
    if splitk > 1:
        wbytes = h.rs_gemm_workspace_bytes(M, N, splitk)
        wbuf = ws.get("gemm_splitk", wbytes // 4)
    if dropout is not None and dropout[0] > 0 and (accumulate or (defer_reduce is not None and splitk > 1)):
        raise RSError("gemm: a dropout mask goes with a plain product only")
    if accumulate:
        if mulsrc is not None or colsum_part is not None:
            raise RSError("gemm: accumulate takes bias and activation only")
        call("rs_gemm_f32_acc", int(a_mc), int(b_nc), M, N, K, ptr(a), lda, ptr(b), ldb, ptr(out), ldc, ptr(bias),
             int(act), int(tile), int(splitk), ptr(wbuf), wbytes, stream())
        return out
    part = None
    if colsum_part is not None and splitk == 1:
        aligned = all(t.data_ptr() % 16 == 0 and ld % 4 == 0 for t, ld in ((a, lda), (b, ldb)))
        if tile in (0, 1) and not aligned:
            tile = 3
        rows = h.rs_gemm_colsum_rows(M, tile)
        part = colsum_part.buffer(rows * N)
        colsum_part.add(part, rows, N, N)
    if dropout is not None and dropout[0] > 0:
        p_, seed, salt, call_no, call_dev = dropout
        call("rs_gemm_f32_drop", int(a_mc), int(b_nc), M, N, K, ptr(a), lda, ptr(b), ldb, ptr(out), ldc,
             ptr(bias), int(act), ptr(mulsrc), ldm, int(mul_act), int(tile), int(splitk), ptr(wbuf), wbytes, ptr(part),
             float(p_), int(seed) & 0xFFFFFFFFFFFFFFFF, int(salt) & 0xFFFFFFFF, int(call_no), ptr(call_dev), stream())
    else:
        call("rs_gemm_f32", int(a_mc), int(b_nc), M, N, K, ptr(a), lda, ptr(b), ldb, ptr(out), ldc,
             ptr(bias), int(act), ptr(mulsrc), ldm, int(mul_act), int(tile), int(splitk), ptr(wbuf), wbytes, ptr(part),
             stream())
    if colsum_part is not None and part is None:
        colsum(out, out=colsum_part.take_destination(), ws=ws)        # split reduction: separate column-sum launch
    return out


# LDS ring stages of grouped launches (GemmGroup) in the training steps; 0 = one launch per product.  Measured on the
# DeepFM bench step (B=4096, MLP 624-1024-512-256): 0.396 ms ungrouped, 0.384 with 2 stages, 0.375 with 3, 0.390 with 4.
GROUP_STAGES = int(os.environ.get("RS_GROUP_GEMMS", "3"))


def group_sort_max_entries() -> int:
    """Largest batch whose index sort can ride in a grouped GEMM launch (GemmGroup.launch(sort=...))."""
    return _lib.lib().rs_group_sort_max_entries()


# Radix passes of the one-workgroup index sort that run in the catch-up's launch when the sort is cut in two (the rest
# runs in the gather's launch); 0 = the whole sort beside the catch-up.  Measured at B=4096, 39 fields (rocprofv3): whole
# sort beside the catch-up 27.7 us + gather 9.2 us; 3 passes 19.2 + 17.1; every pass of the 20-bit fields first and only
# the segmentation beside the gather 24.2 + 11.2 -- a sort launch has about 9 us of fixed cost, so the cut buys 1-2 us.
SORT_SPLIT_PASSES = int(os.environ.get("RS_SORT_SPLIT", "4"))


class GemmGroup:
    """Products that may run in one launch (rs_gemm_f32_group): they wait for the same predecessor and nobody reads
    their results before the launch after them -- a layer's dX and dW.  add() takes the arguments of gemm() (64x64
    tiles; no accumulate); launch() issues everything collected, in groups of four.  A product the grouped
    kernel cannot take (operands that do not allow 16-byte accesses) is issued by gemm() right away."""

    def __init__(self, stages: int = 2):
        self.stages = stages
        self._descs = []
        self._keep = []

    def __len__(self):
        return len(self._descs)

    def add(self, a: torch.Tensor, b: torch.Tensor, *, a_mc=False, b_nc=False, bias=None, act=0, mulsrc=None, mul_act=0,
            out=None, splitk=0, colsum_part=None, defer_reduce=None, dropout=None) -> torch.Tensor:
        lda = _chk_rows(a, "a")
        ldb = _chk_rows(b, "b")
        M, K = (a.shape[1], a.shape[0]) if a_mc else (a.shape[0], a.shape[1])
        N, Kb = (b.shape[1], b.shape[0]) if b_nc else (b.shape[0], b.shape[1])
        if K != Kb:
            raise RSError(f"gemm: inner dimensions differ ({K} vs {Kb})")
        _chk(bias, torch.float32, "bias", (N,))
        ldm = _chk_rows(mulsrc, "mulsrc", (M, N))
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        ldc = _chk_rows(out, "out", (M, N))
        if M == 0 or N == 0:
            return out
        h = _lib.lib()
        if splitk <= 0:
            plan = _gemm_plans.get((M, N, K))
            if plan is None:
                import ctypes as C
                sk = C.c_int(1)
                t = h.rs_gemm_plan(M, N, K, 256, C.byref(sk))
                plan = _gemm_plans[(M, N, K)] = (t, sk.value)
            splitk = plan[1]
        splitk = max(1, min(splitk, K // 32))
        ok = all(t.data_ptr() % 16 == 0 and ld % 4 == 0 for t, ld in ((a, lda), (b, ldb)))
        ok = ok and (M % 4 == 0 if a_mc else K % 4 == 0) and (N % 4 == 0 if b_nc else K % 4 == 0)
        split_ok = (defer_reduce is not None and bias is None and act == 0 and mulsrc is None and colsum_part is None
                    and out.is_contiguous())
        has_drop = dropout is not None and dropout[0] > 0
        if not ok or (splitk > 1 and (not split_ok or has_drop)):
            return gemm(a, b, a_mc=a_mc, b_nc=b_nc, bias=bias, act=act, mulsrc=mulsrc, mul_act=mul_act, out=out,
                        colsum_part=colsum_part, defer_reduce=defer_reduce, dropout=dropout)
        dsc = _lib.GemmDesc(a_mc=int(a_mc), b_nc=int(b_nc), M=M, N=N, K=K, A=a.data_ptr(), lda=lda, B=b.data_ptr(), ldb=ldb,
                            splitk=splitk)
        if splitk > 1:
            wbytes = h.rs_gemm_workspace_bytes(M, N, splitk)
            wbuf = defer_reduce.buffer(wbytes // 4)
            defer_reduce.add(wbuf, out, M * N, splitk)
            dsc.workspace, dsc.workspace_bytes = wbuf.data_ptr(), wbytes
            self._keep.append(wbuf)
        else:
            dsc.C, dsc.ldc = out.data_ptr(), ldc
            dsc.bias, dsc.act = (bias.data_ptr() if bias is not None else None), int(act)
            dsc.mulsrc, dsc.ldm, dsc.mul_act = (mulsrc.data_ptr() if mulsrc is not None else None), ldm, int(mul_act)
            if colsum_part is not None:
                rows = h.rs_gemm_colsum_rows(M, 3)
                part = colsum_part.buffer(rows * N)
                colsum_part.add(part, rows, N, N)
                dsc.colsum_part = part.data_ptr()
                self._keep.append(part)
            if has_drop:
                p_, seed, salt, call_no, call_dev = dropout
                dsc.drop_p, dsc.drop_salt = float(p_), int(salt) & 0xFFFFFFFF
                dsc.drop_seed, dsc.drop_call_host = int(seed) & 0xFFFFFFFFFFFFFFFF, int(call_no)
                dsc.drop_call_dev = call_dev.data_ptr() if call_dev is not None else None
                self._keep.append(call_dev)
        self._descs.append(dsc)
        self._keep += [a, b, out, bias, mulsrc]
        return out

    def launch(self, sort=None):
        """sort = (plan, idx, phase): the launch also sorts the columnar batch idx [F, B] into `plan` (SparsePlan.
        sort_columnar of the NEXT batch riding beside these products; needs 3 stages and B <= group_sort_max_entries()).
        phase 0: the whole sort; 1 / 2: its two halves, in two launches (plan.ride_split_passes(B) passes first)."""
        descs, self._descs, self._keep = self._descs, [], []
        for i in range(0, len(descs), 4):
            grp = descs[i:i + 4]
            arr = (_lib.GemmDesc * len(grp))(*grp)
            if sort is not None and i + 4 >= len(descs) and self.stages == 3 and sort[1].shape[1] <= group_sort_max_entries():
                plan, idx, phase = sort
                plan._sort_in_group(arr, len(grp), idx, phase)
                sort = None
            else:
                call("rs_gemm_f32_group", arr, len(grp), int(self.stages), stream())
        if sort is not None:                     # nothing was collected: the sort (or this half of it) in a launch of its own
            sort[0]._sort_in_group(None, 0, sort[1], sort[2])


class SplitKParts:
    """Collects split-K products whose slabs wait in their own workspace buffers and reduces them in one launch.

    Usage: p = SplitKParts(ws); ops.gemm(..., out=dW, defer_reduce=p); ...; p.finish().  A product whose plan does
    not split is computed directly."""

    def __init__(self, ws: "Workspace", tag: str = ""):
        self.ws = ws
        self.tag = tag              # keeps the slab buffers of two collectors that share a workspace apart
        self._items = []            # (slab buffer, out, M*N, splitk)

    def buffer(self, numel: int) -> torch.Tensor:
        return self.ws.get(f"gemm_slabs_{self.tag}{len(self._items)}", numel)

    def add(self, wbuf, out, total: int, splitk: int):
        self._items.append((wbuf, out, int(total), int(splitk)))

    @staticmethod
    def _arrays(grp):
        import ctypes as C
        k = len(grp)
        W = (C.c_void_p * k)(*[g[0].data_ptr() for g in grp])
        O = (C.c_void_p * k)(*[g[1].data_ptr() for g in grp])
        T = (C.c_longlong * k)(*[g[2] for g in grp])
        S = (C.c_int * k)(*[g[3] for g in grp])
        return W, O, T, S, k

    def finish(self):
        items, self._items = self._items, []
        for i in range(0, len(items), 8):
            call("rs_splitk_reduce_multi", *self._arrays(items[i:i + 8]), stream())

    def hand_over(self):
        """The pending products as the (ws, out, total, splitk, count) argument arrays of one reducing launch, for a
        consumer that runs them inside a launch of its own; anything beyond eight products is reduced here."""
        items, self._items = self._items, []
        for i in range(8, len(items), 8):
            call("rs_splitk_reduce_multi", *self._arrays(items[i:i + 8]), stream())
        return self._arrays(items[:8]) if items else None


class ColsumParts:
    """Collects partial column sums left behind by several producers and adds them up in one launch.

    Usage: p = ColsumParts(ws); p.expect(dest); ops.gemm(..., colsum_part=p); ...; p.finish()."""

    def __init__(self, ws: "Workspace", tag: str = ""):
        self.ws = ws
        self.tag = tag              # tells the scratch buffers of collectors that are open at the same time apart
        self._items = []            # (part tensor, rows, n, ld, dest)
        self._dest = None
        self._k = 0

    def expect(self, dest: torch.Tensor):
        """Destination [N] of the next producer's column sums."""
        self._dest = dest
        return self

    def take_destination(self):
        d, self._dest = self._dest, None
        if d is None:
            raise RSError("ColsumParts: expect(dest) must precede the producer")
        return d

    def buffer(self, numel: int) -> torch.Tensor:
        self._k += 1
        return self.ws.get(f"colsum_part{self.tag}_{self._k}", numel)

    def add(self, part: torch.Tensor, rows: int, n: int, ld: int, dest: Optional[torch.Tensor] = None, offset: int = 0):
        dest = dest if dest is not None else self.take_destination()
        _chk(dest, torch.float32, "colsum destination", (n,))
        self._items.append((part, int(rows), int(n), int(ld), dest, int(offset)))

    @staticmethod
    def _arrays(grp):
        import ctypes as C
        k = len(grp)
        P = (C.c_void_p * k)(*[g[0].data_ptr() + 4 * g[5] for g in grp])
        O = (C.c_void_p * k)(*[g[4].data_ptr() for g in grp])
        R = (C.c_int * k)(*[g[1] for g in grp])
        Nn = (C.c_int * k)(*[g[2] for g in grp])
        L = (C.c_int * k)(*[g[3] for g in grp])
        return P, R, Nn, L, O, k

    def finish(self):
        items, self._items, self._k = self._items, [], 0
        for i in range(0, len(items), 8):
            call("rs_colsum_finish_multi", *self._arrays(items[i:i + 8]), stream())

    def hand_over(self):
        """The pending problems as the (part, rows, n, ld, out, count) argument arrays of one finishing launch, for a
        consumer that runs them inside a launch of its own; anything beyond eight problems is finished here."""
        items, self._items, self._k = self._items, [], 0
        for i in range(8, len(items), 8):
            call("rs_colsum_finish_multi", *self._arrays(items[i:i + 8]), stream())
        return self._arrays(items[:8]) if items else None


def colsum(x: torch.Tensor, out=None, ws: Optional[Workspace] = None, scale=None, parts: "ColsumParts" = None) -> torch.Tensor:
    """out[n] = sum_m scale[m] * x[m, n] (scale optional).  parts: a ColsumParts collector whose finish() the caller
    runs anyway -- only the partial sums are formed here and `out` is valid after that finish()."""
    _chk(x, torch.float32, "x")
    M, N = x.shape
    _chk(scale, torch.float32, "scale", (M,))
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
    if parts is not None and M > 0 and N > 0:
        rows = _lib.lib().rs_colsum_partial_rows(M)
        part = parts.buffer(rows * N)
        call("rs_colsum_partials", ptr(x), M, N, _chk_rows(x, "x"), ptr(scale), ptr(part), stream())
        parts.add(part, rows, N, N, dest=out)
        return out
    ws = ws or default_workspace(x.device)
    wbytes = _lib.lib().rs_colsum_workspace_bytes(M, N)
    wbuf = ws.get("colsum", max(wbytes // 4, 1))
    call("rs_colsum_f32", ptr(x), M, N, N, ptr(scale), ptr(out), ptr(wbuf), wbytes, stream())
    return out


# ---------------------------------------------------------------------------------------------
# loss head
# ---------------------------------------------------------------------------------------------
def bce_head(parts, labels: torch.Tensor, want_logit=False, want_prob=False, want_dz=True, grad_scale=1.0,
             sum_dz_out=None, hlast=None, act_last=0, want_dzh=False, sum_dzh_out=None):
    """loss = BCE(sigmoid(sum(parts)), labels) (mean); returns dict(loss, dz, logit, prob, dzh).

    sum_dz_out: 1-element tensor receiving sum(dz) (gradient of a bias added to the logit).
    hlast/act_last: output and activation of a 1-wide last layer that is one of the parts; then
    dzh = dz * act'(hlast) is returned and sum_dzh_out (1 element) receives its sum."""
    parts = [p.reshape(-1) for p in parts if p is not None]
    B = parts[0].numel()
    for p in parts:
        _chk(p, torch.float32, "logit part", (B,))
    if len(parts) > 3:
        raise RSError("bce_head: at most three logit parts")
    parts = parts + [None] * (3 - len(parts))
    dev = parts[0].device
    lf = li = None
    if labels.dtype == torch.float32:
        lf = labels
    elif labels.dtype == torch.int64:
        li = labels
    else:
        raise RSError(f"bce_head: labels must be float32 or int64, got {labels.dtype}")
    _chk(labels, labels.dtype, "labels", (B,))
    _chk(hlast, torch.float32, "hlast")
    out = {"loss": torch.empty(1, dtype=torch.float32, device=dev)}
    if want_logit:
        out["logit"] = torch.empty(B, dtype=torch.float32, device=dev)
    if want_prob:
        out["prob"] = torch.empty(B, dtype=torch.float32, device=dev)
    if want_dz:
        out["dz"] = torch.empty(B, dtype=torch.float32, device=dev)
    if want_dzh:
        out["dzh"] = torch.empty(B, dtype=torch.float32, device=dev)
    call("rs_bce_head", ptr(parts[0]), ptr(parts[1]), ptr(parts[2]), ptr(lf), ptr(li), B, ptr(out.get("logit")),
         ptr(out.get("prob")), ptr(out["loss"]), ptr(out.get("dz")), float(grad_scale), ptr(sum_dz_out),
         ptr(hlast.reshape(-1) if hlast is not None else None), int(act_last), ptr(out.get("dzh")), ptr(sum_dzh_out),
         stream())
    return out


def head_fwd_bce(x: torch.Tensor, w: torch.Tensor, bias, act: int, parts, labels: torch.Tensor, grad_scale=1.0,
                 sum_dz_out=None, sum_dzh_out=None, ws: Optional[Workspace] = None, defer: "ColsumParts" = None):
    """1-wide last layer + loss head in one launch: h = act(x.w + b); logit = sum(parts) + h.
    Returns dict(loss [1], dz [B], dzh [B], h [B]).  defer: a ColsumParts collector; the loss and the two batch sums
    are then only valid after its finish() (the kernel leaves per-workgroup partial sums and skips its own tail)."""
    _chk(x, torch.float32, "x")
    B, K = x.shape
    _chk(w, torch.float32, "w")
    if w.numel() != K:
        raise RSError("head_fwd_bce: weight length differs from the input width")
    parts = [p.reshape(-1) for p in parts if p is not None]
    if len(parts) > 2:
        raise RSError("head_fwd_bce: at most two other logit parts")
    for p in parts:
        _chk(p, torch.float32, "logit part", (B,))
    parts = parts + [None] * (2 - len(parts))
    lf = li = None
    if labels.dtype == torch.float32:
        lf = labels
    elif labels.dtype == torch.int64:
        li = labels
    else:
        raise RSError(f"head_fwd_bce: labels must be float32 or int64, got {labels.dtype}")
    _chk(labels, labels.dtype, "labels", (B,))
    dev = x.device
    ws = ws or default_workspace(dev)
    wbuf = ws.get("head_ws", _lib.lib().rs_head_workspace_floats(B, K))
    sem = ws.get("head_sem", 1, dtype=torch.int32, zero=True) if defer is None else None
    out = {k: torch.empty(B, dtype=torch.float32, device=dev) for k in ("dz", "dzh", "h")}
    out["loss"] = torch.empty(1, dtype=torch.float32, device=dev)
    call("rs_head_fwd_bce", ptr(x), B, K, ptr(w), ptr(bias), int(act), ptr(parts[0]), ptr(parts[1]), ptr(lf), ptr(li),
         float(grad_scale), ptr(out["h"]), ptr(out["loss"]), ptr(out["dz"]), ptr(sum_dz_out), ptr(out["dzh"]),
         ptr(sum_dzh_out), ptr(wbuf), ptr(sem), stream())
    if defer is not None:
        rows = (B + 15) // 16
        defer.add(wbuf, rows, 1, 4, dest=out["loss"], offset=0)
        if sum_dz_out is not None:
            defer.add(wbuf, rows, 1, 4, dest=sum_dz_out, offset=1)
        if sum_dzh_out is not None:
            defer.add(wbuf, rows, 1, 4, dest=sum_dzh_out, offset=2)
    return out


def head_fwd_bwd(x: torch.Tensor, w: torch.Tensor, bias, act: int, parts, labels: torch.Tensor, defer: "ColsumParts",
                 dW_out: torch.Tensor, grad_scale=1.0, sum_dz_out=None, sum_dzh_out=None, mul_act: int = 0, dbp_out=None,
                 ws: Optional[Workspace] = None):
    """head_fwd_bce(..., defer=defer) and head_bwd(...) in one launch (x is read once; x.shape[1] <= 1024).
    Returns dict(loss [1], dz [B], dzh [B], h [B], dx [B, K]); loss, the batch sums, dW_out and dbp_out are valid
    after defer.finish()."""
    _chk(x, torch.float32, "x")
    B, K = x.shape
    _chk(w, torch.float32, "w")
    if w.numel() != K:
        raise RSError("head_fwd_bwd: weight length differs from the input width")
    parts = [p.reshape(-1) for p in parts if p is not None]
    if len(parts) > 2:
        raise RSError("head_fwd_bwd: at most two other logit parts")
    for p in parts:
        _chk(p, torch.float32, "logit part", (B,))
    parts = parts + [None] * (2 - len(parts))
    lf = li = None
    if labels.dtype == torch.float32:
        lf = labels
    elif labels.dtype == torch.int64:
        li = labels
    else:
        raise RSError(f"head_fwd_bwd: labels must be float32 or int64, got {labels.dtype}")
    _chk(labels, labels.dtype, "labels", (B,))
    dev = x.device
    ws = ws or default_workspace(dev)
    h = _lib.lib()
    wbuf = ws.get("head_ws", h.rs_head_workspace_floats(B, K))
    rows = h.rs_head_bwd_rows(B)
    part = defer.buffer(rows * 2 * K)
    out = {k: torch.empty(B, dtype=torch.float32, device=dev) for k in ("dz", "dzh", "h")}
    out["loss"] = torch.empty(1, dtype=torch.float32, device=dev)
    out["dx"] = torch.empty_like(x)
    call("rs_head_fwd_bwd", ptr(x), B, K, ptr(w), ptr(bias), int(act), ptr(parts[0]), ptr(parts[1]), ptr(lf), ptr(li),
         float(grad_scale), ptr(out["h"]), ptr(out["dz"]), ptr(out["dzh"]), ptr(wbuf), int(mul_act), ptr(out["dx"]),
         ptr(part), stream())
    defer.add(wbuf, rows, 1, 4, dest=out["loss"], offset=0)
    if sum_dz_out is not None:
        defer.add(wbuf, rows, 1, 4, dest=sum_dz_out, offset=1)
    if sum_dzh_out is not None:
        defer.add(wbuf, rows, 1, 4, dest=sum_dzh_out, offset=2)
    defer.add(part, rows, K, 2 * K, dest=dW_out, offset=0)
    if dbp_out is not None:
        defer.add(part, rows, K, 2 * K, dest=dbp_out, offset=K)
    return out


def head_bwd(dzh: torch.Tensor, w: torch.Tensor, x: torch.Tensor, mul_act: int, parts: "ColsumParts",
             dW_out: torch.Tensor, dbp_out=None) -> torch.Tensor:
    """Backward of the 1-wide layer; returns dx and registers the partial sums of dW_out [K] (and of dbp_out [K] =
    column sums of dx, the previous block's bias gradient when its act' is fused through mul_act) with `parts`."""
    M, K = x.shape
    _chk(dzh, torch.float32, "dzh", (M,))
    rows = _lib.lib().rs_head_bwd_rows(M)
    part = parts.buffer(rows * 2 * K)
    dx = torch.empty_like(x)
    call("rs_head_bwd", ptr(dzh), ptr(w), ptr(x), int(mul_act), M, K, ptr(dx), ptr(part), stream())
    parts.add(part, rows, K, 2 * K, dest=dW_out, offset=0)
    if dbp_out is not None:
        parts.add(part, rows, K, 2 * K, dest=dbp_out, offset=K)
    return dx


def rowdot_fwd(x: torch.Tensor, w: torch.Tensor, bias, act: int) -> torch.Tensor:
    """h[m] = act(x[m, :] . w + bias): a 1-wide linear layer."""
    _chk(x, torch.float32, "x")
    M, K = x.shape
    _chk(w, torch.float32, "w")
    if w.numel() != K:
        raise RSError("rowdot_fwd: weight length differs from the input width")
    h = torch.empty(M, dtype=torch.float32, device=x.device)
    call("rs_rowdot_fwd", ptr(x), M, K, ptr(w), ptr(bias), int(act), ptr(h), stream())
    return h


def rowdot2_fwd(x1: torch.Tensor, x2: torch.Tensor, w: torch.Tensor, bias, act: int) -> torch.Tensor:
    """h[m] = act(cat([x1, x2], 1)[m, :] . w + bias) without the concatenation."""
    _chk(x1, torch.float32, "x1")
    _chk(x2, torch.float32, "x2", (x1.shape[0], x2.shape[1]))
    M, K1, K2 = x1.shape[0], x1.shape[1], x2.shape[1]
    _chk(w, torch.float32, "w")
    if w.numel() != K1 + K2:
        raise RSError("rowdot2_fwd: weight length differs from the input width")
    h = torch.empty(M, dtype=torch.float32, device=x1.device)
    call("rs_rowdot2_fwd", ptr(x1), K1, ptr(x2), K2, M, ptr(w), ptr(bias), int(act), ptr(h), stream())
    return h


def head2_bwd(dzh: torch.Tensor, w: torch.Tensor, x1: torch.Tensor, x2: torch.Tensor, mul_act2: int, parts: "ColsumParts",
              dW_out: torch.Tensor, dbp_out=None):
    """Backward of rowdot2_fwd: returns (dx1, dx2) and registers the partial sums of dW_out [K1 + K2] (and of dbp_out
    [K2] = column sums of dx2, the bias gradient of the block whose output x2 is when its act' is fused through
    mul_act2) with `parts`."""
    M, K1, K2 = x1.shape[0], x1.shape[1], x2.shape[1]
    K = K1 + K2
    _chk(dzh, torch.float32, "dzh", (M,))
    rows = _lib.lib().rs_head_bwd_rows(M)
    part = parts.buffer(rows * 2 * K)
    dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
    call("rs_head2_bwd", ptr(dzh), ptr(w), ptr(x1), K1, ptr(x2), K2, int(mul_act2), M, ptr(dx1), ptr(dx2), ptr(part),
         stream())
    parts.add(part, rows, K, 2 * K, dest=dW_out, offset=0)
    if dbp_out is not None:
        parts.add(part, rows, K2, 2 * K, dest=dbp_out, offset=K + K1)
    return dx1, dx2


def rowdot_bwd_dx(dz: torch.Tensor, w: torch.Tensor, x: torch.Tensor, mul_act: int) -> torch.Tensor:
    """dx[m, k] = dz[m] * w[k] * act'(x[m, k])."""
    M, K = x.shape
    _chk(dz, torch.float32, "dz", (M,))
    dx = torch.empty_like(x)
    call("rs_rowdot_bwd_dx", ptr(dz), ptr(w), ptr(x), int(mul_act), M, K, ptr(dx), stream())
    return dx


# ---------------------------------------------------------------------------------------------
# sparse backward
# ---------------------------------------------------------------------------------------------
class SparsePlan:
    """Device scratch of one sparse backward: sort/segment plan + reduction buffers.

    cap = max entries per field.  Buffers are allocated once and reused every step.
    """


    def __init__(self, layout: FieldLayout, cap: int, d: int, with_lr: bool, device=None):
        h = _lib.lib()
        self.layout = layout
        self.F = layout.num_fields
        self.cap = int(cap)
        self.d = int(d)
        dev = device or layout.device
        if cap > h.rs_merge_max_entries():      # the sort itself takes rs_sort_max_entries(), merge_lists this many
            raise RSError(f"sparse backward: {cap} entries per field exceed the plan limit {h.rs_merge_max_entries()}")
        # entries per wave window of the segmented reduction (chunks are cut at its multiples)
        self.CHUNK = h.rs_sparse_window(self.d)
        if self.CHUNK <= 0:
            raise RSError(f"sparse backward: unsupported embedding_dim {d}")
        self.plan = torch.zeros(h.rs_sort_plan_ints(self.F, self.cap), dtype=torch.int32, device=dev)
        self.chunk_sum = torch.empty((self.F, self.cap, self.d), dtype=torch.float32, device=dev)
        self.rowgrad = torch.empty((self.F, self.cap, self.d), dtype=torch.float32, device=dev)
        self.chunk_sum1 = torch.empty((self.F, self.cap), dtype=torch.float32, device=dev) if with_lr else None
        self.rowgrad1 = torch.empty((self.F, self.cap), dtype=torch.float32, device=dev) if with_lr else None
        self.n_partials = h.rs_sparse_bwd_num_partials(self.F, self.cap, self.d)
        if self.n_partials < 0:
            raise RSError(f"sparse backward: unsupported embedding_dim {d}")
        self._columnar = {}

    # views into the plan (for tests / inspection)
    def counts(self):
        off = self.F * (6 * self.cap + 2)
        return self.plan[off:off + 4 * self.F].view(self.F, 4)

    def seg_rows(self):
        off = 3 * self.F * self.cap + self.F * (self.cap + 1)
        return self.plan[off:off + self.F * self.cap].view(self.F, self.cap)

    def check(self):
        """Raise if merge_lists was handed more entries for a field than it was sized for (device sync): such a field
        gets an empty plan, i.e. its gradients are dropped."""
        if bool((self.counts()[:, 3] < 0).any().item()):
            raise RSError("sparse backward: a field's key lists held more entries than the merge was sized for")

    def columnar_lists(self, B: int):
        """Sub-list descriptors for a columnar [F, B] batch: one list per field."""
        if B not in self._columnar:
            dev = self.plan.device
            f = torch.arange(self.F, dtype=torch.int32, device=dev)
            self._columnar[B] = (f * B, torch.full((self.F,), B, dtype=torch.int32, device=dev), f.clone())
        return self._columnar[B]

    def sort_columnar(self, idx: torch.Tensor):
        """idx [F, B] int64 -> plan.  The gradient-row index of entry (f, b) is b*F + f."""
        Fn, B = idx.shape
        _chk(idx, torch.int64, "idx", (self.F, B))
        if B > self.cap:
            raise RSError(f"sparse backward: batch {B} exceeds plan capacity {self.cap}")
        off, cnt, base = self.columnar_lists(B)
        call("rs_sparse_bwd_sort", ptr(idx), ptr(off), ptr(cnt), ptr(base), self.F, 1, self.F, self.cap, B,
             self.CHUNK, ptr(self.layout.vocab_t), ptr(self.layout.row_offset_t), ptr(self.plan), stream())

    @staticmethod
    def ride_split_passes(B: int) -> int:
        """Radix passes in the first half of a sort that rides in two GEMM launches (4 key bits per pass above 1024
        entries per field, 6 below): three of the five passes of a 20-bit key, two of the four of an 18..24-bit one."""
        return 3 if B > 1024 else 2

    def _sort_in_group(self, descs, count: int, idx: torch.Tensor, phase: int = 0):
        Fn, B = idx.shape
        _chk(idx, torch.int64, "idx", (self.F, B))
        if B > self.cap:
            raise RSError(f"sparse backward: batch {B} exceeds plan capacity {self.cap}")
        off, cnt, base = self.columnar_lists(B)
        call("rs_gemm_f32_group_sort", descs, int(count), ptr(idx), ptr(off), ptr(cnt), ptr(base), self.cap, self.CHUNK,
             ptr(self.layout.vocab_t), ptr(self.layout.row_offset_t), ptr(self.plan), B, Fn, int(phase),
             self.ride_split_passes(B) if phase else 0, stream())

    def catchup_plan(self, main, one, mode: int = 0):
        """Exact Adam on touched rows for the DISTINCT rows of the batch this plan has sorted (main / one: the RowAdam
        states of the [R, d] and the 1-dim table).  mode 0: rows behind the step count are replayed up to it and every
        row of the plan is stamped step + 1 -- promised to the update that steps it next, which must then write no stamps
        itself; mode 1 takes that promise back; mode 2 is the plain replay (rows behind are brought up to the step count
        and stamped with it; the update that follows stamps as usual)."""
        ref = main or one
        if ref is None:
            return
        for st in (main, one):
            if st is not None and st.rows != self.layout.total_rows:
                raise RSError("catchup_plan: table and field layout disagree on the number of rows")
        g = lambda st, k: ptr(getattr(st, k)) if st is not None else None
        call("rs_adam_catchup_plan", ptr(self.plan), self.F, self.cap, main.d if main else 1, g(main, "w"), g(main, "m"),
             g(main, "v"), g(main, "last"), g(one, "w"), g(one, "m"), g(one, "v"), g(one, "last"), OPT_CODES[ref.opt],
             ptr(ref.step), ptr(ref.hyper), ptr(ref.bctab), ref.bctab.shape[0], ref.kmax, int(mode), stream())

    def catchup_serve(self, main, one, pay: torch.Tensor, ld_pay: int, col: int, col1: int):
        """catchup_plan(mode 2) that also writes every row of the plan, once up to date, to the positions that name it
        (the plan's entries) of `pay` (flat, ld_pay floats per position; d-wide row at column col, 1-dim value at col1)."""
        ref = main or one
        if ref is None:
            raise RSError("catchup_serve: needs a RowAdam state")
        for st in (main, one):
            if st is not None and st.rows != self.layout.total_rows:
                raise RSError("catchup_serve: table and field layout disagree on the number of rows")
        _chk(pay, torch.float32, "pay")
        g = lambda st, k: ptr(getattr(st, k)) if st is not None else None
        call("rs_adam_catchup_serve", ptr(self.plan), self.F, self.cap, main.d if main else 1, g(main, "w"), g(main, "m"),
             g(main, "v"), g(main, "last"), g(one, "w"), g(one, "m"), g(one, "v"), g(one, "last"), OPT_CODES[ref.opt],
             ptr(ref.step), ptr(ref.hyper), ptr(ref.bctab), ref.bctab.shape[0], ref.kmax, ptr(pay), int(ld_pay), int(col),
             int(col1), stream())

    def update_with_dense_ahead(self, opt: str, hyper, scal, dense_w, dense_g, dense_m, dense_v, next_plan: "SparsePlan",
                                table=None, m=None, v=None, table1=None, m1=None, v1=None, rs=None, rs1=None):
        """update_with_dense (no stamps written: the rows of this plan were promised to this update by catchup_plan) and
        next_plan.catchup_plan(rs, rs1) in one launch."""
        _chk(dense_w, torch.float32, "dense_w")
        _chk(dense_g, torch.float32, "dense_g", dense_w.shape)
        ref = rs or rs1
        if ref is None or next_plan.cap != self.cap or next_plan.F != self.F:
            raise RSError("update_with_dense_ahead: needs the tables' RowAdam states and a next plan of the same shape")
        call("rs_sparse_dense_update_ahead", ptr(self.plan), self.F, self.cap, self.d, OPT_CODES[opt], ptr(self.rowgrad),
             ptr(self.rowgrad1), ptr(table), ptr(m), ptr(v), ptr(table1), ptr(m1), ptr(v1), ptr(dense_w), ptr(dense_g),
             ptr(dense_m), ptr(dense_v), dense_w.numel(), ptr(hyper), ptr(scal), ptr(next_plan.plan),
             ptr(rs.last) if rs is not None else None, ptr(rs1.last) if rs1 is not None else None, ptr(ref.step),
             ptr(ref.bctab), ref.bctab.shape[0], ref.kmax, stream())

    def sort_and_gather(self, idx: torch.Tensor, table, table1, bias, want_emb=True, want_sum=False, want_lr=False,
                        want_fm=False, resume_passes: int = 0):
        """sort_columnar(idx) and gather_lr_fm_fwd(idx, ...) in one launch; returns the gather's dict.
        resume_passes > 0: the sort was begun by sort_and_catchup(..., split_passes=resume_passes) and is finished here."""
        Fn, B = idx.shape
        _chk(idx, torch.int64, "idx", (self.F, B))
        if B > self.cap:
            raise RSError(f"sparse backward: batch {B} exceeds plan capacity {self.cap}")
        layout = self.layout
        d = 1
        if table is not None:
            d = table.shape[1]
            _chk(table, torch.float32, "table", (layout.total_rows, d))
        if table1 is not None:
            if table1.dim() == 2 and table1.shape[1] == 1:
                table1 = table1.view(-1)
            _chk(table1, torch.float32, "table1", (layout.total_rows,))
        _chk(bias, torch.float32, "bias", (1,))
        dev = idx.device
        out = {}
        if want_emb:
            out["emb"] = torch.empty((B, Fn, d), dtype=torch.float32, device=dev)
        if want_sum:
            out["sum"] = torch.empty((B, d), dtype=torch.float32, device=dev)
        if want_lr:
            out["lr"] = torch.empty((B,), dtype=torch.float32, device=dev)
        if want_fm:
            out["fm"] = torch.empty((B,), dtype=torch.float32, device=dev)
        off, cnt, base = self.columnar_lists(B)
        call("rs_sort_gather", ptr(idx), ptr(off), ptr(cnt), ptr(base), self.cap, self.CHUNK, ptr(layout.vocab_t),
             ptr(layout.row_offset_t), ptr(self.plan), ptr(table), ptr(table1), ptr(bias), ptr(out.get("emb")),
             ptr(out.get("sum")), ptr(None), ptr(out.get("lr")), ptr(out.get("fm")), B, Fn, d, ptr(layout.err_flag),
             int(resume_passes), stream())
        return out

    def split_passes_for(self, B: int) -> int:
        """Radix passes of the index sort that run beside the Adam catch-up when the rest runs beside the gather
        (0: the sort is not cut in two launches for this batch size)."""
        return SORT_SPLIT_PASSES if B <= _lib.lib().rs_sort_split_max_entries() else 0

    def sort_and_catchup(self, idx: torch.Tensor, main, one, split_passes: int = 0):
        """sort_columnar(idx) and adam_catchup(idx, layout, main, one, mark_ahead=0) in one launch (main / one: the
        RowAdam states of the [R, d] and the 1-dim table looked up with idx; the update that follows stamps the rows).
        split_passes > 0: only that many radix passes of the sort run here; sort_and_gather(..., resume_passes=
        split_passes) must follow and writes the plan."""
        Fn, B = idx.shape
        _chk(idx, torch.int64, "idx", (self.F, B))
        if B > self.cap:
            raise RSError(f"sparse backward: batch {B} exceeds plan capacity {self.cap}")
        ref = main or one
        for st in (main, one):
            if st is not None and st.rows != self.layout.total_rows:
                raise RSError("sort_and_catchup: table and field layout disagree on the number of rows")
        if main is not None and one is not None and (main.step is not one.step or main.opt != one.opt):
            raise RSError("sort_and_catchup: the two tables belong to different optimizers")
        if main is not None and main.d > 256 and not split_passes:
            # rows of hundreds of floats: the sort first, then the replay with a wave per DISTINCT row (the entry-wise
            # form elects a row's owner among the entries of a workgroup and replays the elected rows a few at a time)
            self.sort_columnar(idx)
            self.catchup_plan(main, one, mode=2)
            return
        g = lambda st, k: ptr(getattr(st, k)) if st is not None else None
        off, cnt, base = self.columnar_lists(B)
        call("rs_sort_catchup", ptr(idx), ptr(off), ptr(cnt), ptr(base), self.cap, self.CHUNK, ptr(self.layout.vocab_t),
             ptr(self.layout.row_offset_t), ptr(self.plan), B, Fn, main.d if main else 1, g(main, "w"), g(main, "m"),
             g(main, "v"), g(main, "last"), g(one, "w"), g(one, "m"), g(one, "v"), g(one, "last"), OPT_CODES[ref.opt],
             ptr(ref.step), ptr(ref.hyper), ptr(ref.bctab), ref.bctab.shape[0], ref.kmax, int(split_passes), stream())

    def sort_lists(self, keys, sub_off, sub_cnt, pay_base, nsrc: int, max_entries: int):
        """General form: per field, `nsrc` packed sub-lists of keys; gradient rows are packed likewise."""
        _chk(keys, torch.int64, "keys")
        for t, n in ((sub_off, "sub_off"), (sub_cnt, "sub_cnt"), (pay_base, "pay_base")):
            _chk(t, torch.int32, n, (nsrc * self.F,))
        call("rs_sparse_bwd_sort", ptr(keys), ptr(sub_off), ptr(sub_cnt), ptr(pay_base), 1, nsrc, self.F, self.cap,
             int(max_entries), self.CHUNK, ptr(self.layout.vocab_t), ptr(self.layout.row_offset_t), ptr(self.plan),
             stream())

    def merge_lists(self, keys, sub_off, sub_cnt, pay_base, nsrc: int, max_entries: int):
        """sort_lists for sub-lists that are each ascending already: the same plan from a merge instead of a sort."""
        _chk(keys, torch.int64, "keys")
        for t, n in ((sub_off, "sub_off"), (sub_cnt, "sub_cnt"), (pay_base, "pay_base")):
            _chk(t, torch.int32, n, (nsrc * self.F,))
        call("rs_sparse_bwd_merge_lists", ptr(keys), ptr(sub_off), ptr(sub_cnt), ptr(pay_base), 1, nsrc, self.F,
             self.cap, int(max_entries), self.CHUNK, ptr(self.layout.vocab_t), ptr(self.layout.row_offset_t),
             ptr(self.plan), stream())

    def reduce(self, norm_partials: torch.Tensor, gout=None, g1src=None, dfm=None, S=None, emb=None, dlr=None,
               Fdiv=None, dense=None, dense_partials=None, colsum=None, splitk=None, ld_gout: int = 0, ld_g1: int = 0,
               routed=None):
        """Row gradients + ||g||^2 partials (norm_partials: >= n_partials floats).  dense / dense_partials: also
        write the sum-of-squares partials of the flat buffer `dense` (one per element of dense_partials) from the
        same launch.  colsum / splitk: ColsumParts.hand_over() / SplitKParts.hand_over() -- those run beside the first launch.
        routed = (route [F, cap] int32, rows, rows1, ld_row, ld_row1): the gradient of segment (f, u) goes to position
        route[f, u] of rows / rows1 (flat views, ld floats per position) instead of this plan's rowgrad / rowgrad1."""
        if norm_partials.numel() < self.n_partials:
            raise RSError("sparse backward: norm_partials too small")
        # ld_gout / ld_g1: gout / g1src are columns of a wider buffer (floats between consecutive rows / values)
        for t, n in ((gout, "gout"), (g1src, "g1src"), (dfm, "dfm"), (S, "S"), (emb, "emb"), (dlr, "dlr")):
            if t is not None and ((t is gout and ld_gout) or (t is g1src and ld_g1)):
                if t.dtype != torch.float32:
                    raise RSError(f"{n}: expected dtype torch.float32, got {t.dtype}")
                require_cuda(t)
            else:
                _chk(t, torch.float32, n)
        if routed is not None:
            route, rows, rows1, ld_row, ld_row1 = routed
            _chk(route, torch.int32, "route", (self.F, self.cap))
            if (rows1 is None) != (self.rowgrad1 is None):
                raise RSError("sparse backward: routed destinations do not match the plan's tables")
            if dense is not None:
                _chk(dense, torch.float32, "dense")
                _chk(dense_partials, torch.float32, "dense_partials")
            call("rs_sparse_bwd_reduce_routed", ptr(self.plan), self.F, self.cap, self.d, ptr(gout), ptr(g1src), ptr(dfm),
                 ptr(S), ptr(emb), ptr(dlr), int(Fdiv or self.F), ptr(self.chunk_sum), ptr(self.chunk_sum1),
                 ptr(rows), ptr(rows1), ptr(norm_partials), ptr(dense),
                 dense.numel() if dense is not None else 0, ptr(dense_partials if dense is not None else None),
                 dense_partials.numel() if dense is not None else 0, *(colsum or (None, None, None, None, None, 0)),
                 *(splitk or (None, None, None, None, 0)), int(ld_gout), int(ld_g1), ptr(route), int(ld_row), int(ld_row1),
                 stream())
            return
        if dense is not None or colsum is not None or splitk is not None:
            if dense is not None:
                _chk(dense, torch.float32, "dense")
                _chk(dense_partials, torch.float32, "dense_partials")
            call("rs_sparse_bwd_reduce_sumsq", ptr(self.plan), self.F, self.cap, self.d, ptr(gout), ptr(g1src), ptr(dfm),
                 ptr(S), ptr(emb), ptr(dlr), int(Fdiv or self.F), ptr(self.chunk_sum), ptr(self.chunk_sum1),
                 ptr(self.rowgrad), ptr(self.rowgrad1), ptr(norm_partials), ptr(dense),
                 dense.numel() if dense is not None else 0, ptr(dense_partials if dense is not None else None),
                 dense_partials.numel() if dense is not None else 0, *(colsum or (None, None, None, None, None, 0)),
                 *(splitk or (None, None, None, None, 0)), int(ld_gout), int(ld_g1), stream())
            return
        call("rs_sparse_bwd_reduce", ptr(self.plan), self.F, self.cap, self.d, ptr(gout), ptr(g1src), ptr(dfm), ptr(S),
             ptr(emb), ptr(dlr), int(Fdiv or self.F), ptr(self.chunk_sum), ptr(self.chunk_sum1), ptr(self.rowgrad),
             ptr(self.rowgrad1), ptr(norm_partials), int(ld_gout), int(ld_g1), stream())

    @staticmethod
    def _stamps(rs, rs1):
        """(last, last1, step) pointers of the tables' RowAdam states: the update stamps every row it steps."""
        ref = rs or rs1
        return (ptr(rs.last) if rs is not None else None, ptr(rs1.last) if rs1 is not None else None,
                ptr(ref.step) if ref is not None else None)

    def update_with_dense(self, opt: str, hyper, scal, dense_w, dense_g, dense_m, dense_v, table=None, m=None, v=None,
                          table1=None, m1=None, v1=None, rs=None, rs1=None):
        """Touched-rows step and the flat dense step in one launch.  rs / rs1: RowAdam of table / table1."""
        _chk(dense_w, torch.float32, "dense_w")
        _chk(dense_g, torch.float32, "dense_g", dense_w.shape)
        call("rs_sparse_dense_update", ptr(self.plan), self.F, self.cap, self.d, OPT_CODES[opt], ptr(self.rowgrad),
             ptr(self.rowgrad1), ptr(table), ptr(m), ptr(v), ptr(table1), ptr(m1), ptr(v1), ptr(dense_w), ptr(dense_g),
             ptr(dense_m), ptr(dense_v), dense_w.numel(), ptr(hyper), ptr(scal), *self._stamps(rs, rs1), stream())

    def update(self, opt: str, hyper, scal, table=None, m=None, v=None, table1=None, m1=None, v1=None, rs=None, rs1=None):
        call("rs_sparse_update", ptr(self.plan), self.F, self.cap, self.d, OPT_CODES[opt], ptr(self.rowgrad),
             ptr(self.rowgrad1), ptr(table), ptr(m), ptr(v), ptr(table1), ptr(m1), ptr(v1), ptr(hyper), ptr(scal),
             *self._stamps(rs, rs1), stream())

    def scatter_dense(self, grad=None, grad1=None):
        call("rs_sparse_scatter_dense", ptr(self.plan), self.F, self.cap, self.d, ptr(self.rowgrad),
             ptr(self.rowgrad1), ptr(grad), ptr(grad1), stream())

    def touched_dot(self, table=None, table1=None) -> torch.Tensor:
        """Partial sums of <reduced gradient, table row> over the distinct rows of the batch (both tables)."""
        n = self.F * _lib.lib().rs_sparse_touched_dot_groups(self.cap, self.d)
        if getattr(self, "_dot_partials", None) is None:
            self._dot_partials = torch.zeros(n, dtype=torch.float32, device=self.plan.device)
        call("rs_sparse_touched_dot", ptr(self.plan), self.F, self.cap, self.d, ptr(self.rowgrad), ptr(self.rowgrad1),
             ptr(table), ptr(table1), ptr(self._dot_partials), n, stream())
        return self._dot_partials


def sumsq_partials(x: torch.Tensor, partials: torch.Tensor):
    _chk(x, torch.float32, "x")
    call("rs_sumsq_partials", ptr(x), x.numel(), ptr(partials), partials.numel(), stream())


def clip_coef(partials: torch.Tensor, max_norm: float, scal: torch.Tensor):
    call("rs_clip_coef", ptr(partials), partials.numel(), float(max_norm), ptr(scal), stream())


def clip_and_prepare(partials: torch.Tensor, max_norm: float, scal: torch.Tensor, step: torch.Tensor,
                     hyper: torch.Tensor):
    """clip_coef + opt_prepare in one launch."""
    call("rs_clip_and_prepare", ptr(partials), partials.numel(), float(max_norm), ptr(scal), ptr(step), ptr(hyper),
         stream())


class TableShrink:
    """SGD with the reference's L2 term on embedding tables (model/utils/BaseModel.py:214-231: lambda/2 ||W||^2 of every
    nn.Embedding) for a trainer that steps touched rows only.  Every row has the gradient g_r + lambda w_r, so rows off
    the batch are multiplied by s = 1 - lr c lambda in every step (c: the clip coefficient).  The tables are kept as
    w / G with ONE device scalar G for all of them: the shrink is G <- G s, a touched row's step is the plain SGD step
    with c / G' in place of c, lookups multiply by G (gather_lr_fm_fwd(..., scale=gscale)), and the squared norm of
    all rows, S, which the clipped norm needs, is carried along in closed form (rs_clip_and_prepare_l2).

    Attached to every table holder as ``holder.row_state``: whoever reads a table outside the trainer (module forward,
    evaluate, state_dict) calls flush(), which multiplies the stored rows by G and sets G = 1."""

    FLUSH_EVERY = 4096       # steps between flushes, so that G stays far from the float32 underflow range

    def __init__(self, weights: List[torch.Tensor], emb_l2: float, device):
        self.weights = weights
        self.emb_l2 = float(emb_l2)
        self.state = torch.tensor([1.0, 0.0], dtype=torch.float64, device=device)     # G, S
        self.gscale = torch.ones(1, dtype=torch.float32, device=device)
        self.scal_tab = torch.tensor([0.0, 1.0], dtype=torch.float32, device=device)
        self._sq = torch.zeros(len(weights), 256, dtype=torch.float32, device=device)
        self.dirty = False           # the tables hold w / G with G != 1
        self.s_stale = True          # S must be taken from the tables again before the next step
        self.since_flush = 0

    def tensors(self):
        return [self.state, self.gscale, self.scal_tab]

    def clip_and_prepare(self, partials, n_sparse, dot_partials, max_norm, scal, step, hyper, loss=None):
        call("rs_clip_and_prepare_l2", ptr(partials), int(n_sparse), partials.numel(), ptr(dot_partials),
             dot_partials.numel(), float(max_norm), self.emb_l2, ptr(self.state), ptr(scal), ptr(self.scal_tab),
             ptr(self.gscale), ptr(loss), ptr(step), ptr(hyper), stream())

    def stepped(self):
        """Host bookkeeping in front of a step (captured or not).  Whoever asked for a flush since the last step may
        also have written the tables (load_state_dict): S is then summed over them again -- one pass over the tables,
        no host synchronisation -- which also drops the rounding its closed-form updates have gathered."""
        if self.since_flush >= self.FLUSH_EVERY:
            self.flush()
        if self.s_stale:
            for i, w in enumerate(self.weights):
                sumsq_partials(w.view(-1), self._sq[i])
            self.state[1:2].copy_(self._sq.double().sum().view(1))
            self.s_stale = False
        self.dirty = True
        self.since_flush += 1

    def flush(self):
        """Stored rows -> weights (G = 1).  Called by everything that reads or writes the tables outside a step."""
        self.s_stale = True
        if not self.dirty:
            return
        for w in self.weights:
            call("rs_scale_inplace", ptr(w), w.numel(), ptr(self.state), stream())
        self.state[0:1].fill_(1.0)
        self.gscale.fill_(1.0)
        self.dirty = False
        self.since_flush = 0

    def catch_up(self, idx, layout, mark_ahead=0):
        self.flush()


def dense_update(w, g, m, v, opt: str, hyper, scal):
    _chk(w, torch.float32, "w")
    _chk(g, torch.float32, "g", w.shape)
    call("rs_dense_update", ptr(w), ptr(g), ptr(m), ptr(v), w.numel(), OPT_CODES[opt], ptr(hyper), ptr(scal), stream())


def opt_prepare(step: torch.Tensor, hyper: torch.Tensor):
    _chk(step, torch.int64, "step", (1,))
    _chk(hyper, torch.float32, "hyper", (8,))
    call("rs_opt_prepare", ptr(step), ptr(hyper), stream())


def make_hyper(device, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    return torch.tensor([lr, beta1, beta2, eps, weight_decay, lr, 1.0, 0.0], dtype=torch.float32, device=device)


# ---------------------------------------------------------------------------------------------
# dense Adam semantics on touched-rows tables: replay of the skipped zero-gradient steps
# ---------------------------------------------------------------------------------------------
class RowAdam:
    """What rs_adam_catchup / rs_adam_flush need for ONE stacked table that rs_sparse_update steps on touched rows
    only: its weights and moments, `last` (int32 per row: optimizer steps already contained in the row) and the
    trainer's device scalars.  Attached to the table holder as ``holder.row_state`` so that every reader of the
    table (module forward, evaluate, state_dict) first brings the rows it reads up to date."""

    BC_STEPS = 1 << 16

    def __init__(self, w, m, v, opt: str, hyper, step, bctab=None):
        if opt not in ("Adam", "AdamW"):
            raise RSError(f"RowAdam: {opt} has no skipped steps to replay")
        import math
        _, beta1, beta2 = (float(x) for x in hyper[:3].tolist())     # (construction time: one host read)
        self.w, self.m, self.v = w, m, v
        self.opt, self.hyper, self.step = opt, hyper, step
        self.rows = w.shape[0]
        self.d = w.shape[1] if w.dim() == 2 else 1
        self.last = torch.zeros(self.rows, dtype=torch.int32, device=w.device)
        # the terms of a row's drift under zero gradients, m_j / (sqrt(v_j) + eps), decay like (beta1 / sqrt(beta2))^j:
        # beyond kmax replayed steps they are below 1e-12 of the first one and the rest is applied in closed form
        if not (0.0 <= beta1 < 1.0 and 0.0 < beta2 < 1.0):
            raise RSError(f"RowAdam: betas ({beta1}, {beta2}) outside [0, 1)")
        ratio = beta1 / math.sqrt(beta2)
        if ratio >= 1.0:
            raise RSError(f"RowAdam: beta1 = {beta1} >= sqrt(beta2) = {math.sqrt(beta2):.6f}: the drift of an untouched row "
                          "does not die out, so its skipped steps cannot be replayed in bounded time; use exact_adam=False "
                          "or the dense update")
        self.kmax = max(1, int(math.ceil(math.log(1e-12) / math.log(ratio)))) if ratio > 0.0 else 1
        self.bctab = bctab if bctab is not None else adam_bc_table(hyper, self.BC_STEPS)

    def catch_up(self, idx: torch.Tensor, layout: FieldLayout, mark_ahead: int = 0):
        adam_catchup(idx, layout, self if self.d != 1 else None, self if self.d == 1 else None, mark_ahead)

    def flush(self):
        call("rs_adam_flush", ptr(self.w), ptr(self.m), ptr(self.v), ptr(self.last), self.rows, self.d,
             OPT_CODES[self.opt], ptr(self.step), ptr(self.hyper), ptr(self.bctab), self.bctab.shape[0], self.kmax,
             stream())

    def tensors(self):
        return [self.last]


def adam_catchup_lists(keys: torch.Tensor, sub_off: torch.Tensor, sub_cnt: torch.Tensor, per_list: int, layout: FieldLayout,
                       main: Optional["RowAdam"], one: Optional["RowAdam"], mark_ahead: int = 0):
    """adam_catchup for keys that arrive as nsrc * F lists (list s * F + f of field f: keys[sub_off[l] : + sub_cnt[l]])."""
    ref = main or one
    if ref is None:
        return
    _chk(keys, torch.int64, "keys")
    nlists = sub_off.numel()
    _chk(sub_off, torch.int32, "sub_off", (nlists,))
    _chk(sub_cnt, torch.int32, "sub_cnt", (nlists,))
    for st in (main, one):
        if st is not None and st.rows != layout.total_rows:
            raise RSError("adam_catchup_lists: table and field layout disagree on the number of rows")
    g = lambda st, k: ptr(getattr(st, k)) if st is not None else None
    call("rs_adam_catchup_lists", ptr(keys), ptr(sub_off), ptr(sub_cnt), nlists, int(per_list), ptr(layout.row_offset_t),
         ptr(layout.vocab_t), layout.num_fields, main.d if main else 1, g(main, "w"), g(main, "m"), g(main, "v"),
         g(main, "last"), g(one, "w"), g(one, "m"), g(one, "v"), g(one, "last"), OPT_CODES[ref.opt], ptr(ref.step),
         int(mark_ahead), ptr(ref.hyper), ptr(ref.bctab), ref.bctab.shape[0], ref.kmax, stream())


def adam_bc_table(hyper: torch.Tensor, n: int) -> torch.Tensor:
    """[n, 2] fp32: {1/(1-beta1^k), 1/sqrt(1-beta2^k)} for k < n, evaluated in double precision on the device."""
    _chk(hyper, torch.float32, "hyper", (8,))
    tab = torch.empty((n, 2), dtype=torch.float32, device=hyper.device)
    call("rs_adam_bc_table", ptr(tab), n, ptr(hyper), stream())
    return tab


def adam_catchup(idx: torch.Tensor, layout: FieldLayout, main: Optional[RowAdam], one: Optional[RowAdam],
                 mark_ahead: int):
    """Replay the skipped zero-gradient Adam steps of the rows batch `idx` [F, B] is about to read: `main` is the
    state of the [R, d] table, `one` of the [R] (1-dim) table looked up with the same indices; one launch."""
    ref = main or one
    if ref is None:
        return
    Fn, B = idx.shape
    _chk(idx, torch.int64, "idx", (layout.num_fields, B))
    for st in (main, one):
        if st is not None and st.rows != layout.total_rows:
            raise RSError("adam_catchup: table and field layout disagree on the number of rows")
    if main is not None and one is not None and (main.step is not one.step or main.opt != one.opt):
        raise RSError("adam_catchup: the two tables belong to different optimizers")
    g = lambda st, k: ptr(getattr(st, k)) if st is not None else None
    call("rs_adam_catchup", ptr(idx), ptr(layout.row_offset_t), ptr(layout.vocab_t), B, Fn, main.d if main else 1,
         g(main, "w"), g(main, "m"), g(main, "v"), g(main, "last"), g(one, "w"), g(one, "m"), g(one, "v"), g(one, "last"),
         OPT_CODES[ref.opt], ptr(ref.step), int(mark_ahead), ptr(ref.hyper), ptr(ref.bctab), ref.bctab.shape[0],
         ref.kmax, stream())


# ---------------------------------------------------------------------------------------------
# activation / normalisation kernels
# ---------------------------------------------------------------------------------------------
def act_fwd(z: torch.Tensor, act: int, slope=None) -> torch.Tensor:
    _chk(z, torch.float32, "z")
    h = torch.empty_like(z)
    call("rs_act_fwd", ptr(z), ptr(h), z.numel(), int(act), ptr(slope), stream())
    return h


def act_bwd(dh: torch.Tensor, act: int, z=None, h=None, slope=None):
    """dz = dh * act'(.); returns (dz, dslope or None)."""
    _chk(dh, torch.float32, "dh")
    dz = torch.empty_like(dh)
    n = dh.numel()
    parts = None
    if act == ACT_CODES["prelu"]:
        parts = torch.empty(_lib.lib().rs_act_bwd_num_partials(n), dtype=torch.float32, device=dh.device)
    call("rs_act_bwd", ptr(dh), ptr(z), ptr(h), ptr(dz), n, int(act), ptr(slope), ptr(parts), stream())
    return dz, (colsum(parts.view(-1, 1)) if parts is not None else None)


# BatchNorm over a batch that is spread over several GPUs: the communicator of the trainer whose step is running (an
# object with .world, .all_gather_into(out, inp) and .all_reduce(t); None: this process holds the whole batch).
_bn_sync = None


@contextlib.contextmanager
def use_bn_sync(comm):
    """Inside the block every training-mode bn_fwd / bn_bwd takes its statistics over the union of the ranks' batches
    (equal batch sizes on all ranks), as the reference's single-process BatchNorm1d does on the whole batch."""
    global _bn_sync
    prev, _bn_sync = _bn_sync, comm
    try:
        yield
    finally:
        _bn_sync = prev


def combine_batch_stats(means: torch.Tensor, variances: torch.Tensor):
    """(mean, biased variance) of the union of W equally large batches from their [W, N] means and biased variances:
    the mean of the means, and the mean of the variances plus the variance of the means -- every term centred, nothing
    cancels."""
    mean = means.mean(0)
    return mean, variances.mean(0) + (means - mean).pow(2).mean(0)


def _bn_fwd_all_ranks(z, gamma, beta, act, running_mean, running_var, eps, momentum, comm):
    M, N = z.shape
    W = comm.world
    st = torch.empty(2 * N, dtype=torch.float32, device=z.device)
    call("rs_bn_stats", ptr(z), M, N, ptr(st), ptr(st[N:]), stream())
    allst = torch.empty((W, 2 * N), dtype=torch.float32, device=z.device)
    comm.all_gather_into(allst, st)
    mean, var = combine_batch_stats(allst[:, :N], allst[:, N:])
    if running_mean is not None and running_var is not None:
        Mg = M * W
        running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
        running_var.mul_(1.0 - momentum).add_(var, alpha=momentum * (Mg / (Mg - 1) if Mg > 1 else 1.0))
    h = torch.empty_like(z)
    xhat = torch.empty_like(z)
    call("rs_bn_fwd", ptr(z), M, N, ptr(gamma), ptr(beta), float(eps), int(act), 0, float(momentum), ptr(mean), ptr(var),
         None, None, ptr(xhat), ptr(h), stream())
    return h, xhat, var


def bn_fwd(z, gamma, beta, act: int, training: bool, running_mean, running_var, eps=1e-5, momentum=0.1):
    """Returns (h, xhat, var_used).  In eval mode the running statistics are used."""
    _chk(z, torch.float32, "z")
    if training and _bn_sync is not None:
        return _bn_fwd_all_ranks(z, gamma, beta, act, running_mean, running_var, eps, momentum, _bn_sync)
    M, N = z.shape
    h = torch.empty_like(z)
    xhat = torch.empty_like(z)
    if training:
        mean = torch.empty(N, dtype=torch.float32, device=z.device)
        var = torch.empty(N, dtype=torch.float32, device=z.device)
    else:
        mean, var = running_mean, running_var
    call("rs_bn_fwd", ptr(z), M, N, ptr(gamma), ptr(beta), float(eps), int(act), int(training), float(momentum),
         ptr(mean), ptr(var), ptr(running_mean) if training else None, ptr(running_var) if training else None,
         ptr(xhat), ptr(h), stream())
    return h, xhat, var


def bn_bwd(dh, h, xhat, act: int, gamma, var, eps=1e-5):
    M, N = dh.shape
    dz = torch.empty_like(dh)
    if _bn_sync is not None:
        # dgamma / dbeta of THIS rank's rows are its share of the parameter gradients (the trainer's all-reduce of the
        # dense gradient sums the shares); dz needs the sums over the union batch
        comm = _bn_sync
        loc = torch.empty(2 * N, dtype=torch.float32, device=dh.device)
        call("rs_bn_bwd_reduce", ptr(dh), ptr(h), ptr(xhat), M, N, int(act), ptr(loc), ptr(loc[N:]), stream())
        tot = loc.clone()
        comm.all_reduce(tot)
        call("rs_bn_bwd_apply", ptr(dh), ptr(h), ptr(xhat), M, N, int(act), ptr(gamma), ptr(var), float(eps), ptr(tot),
             ptr(tot[N:]), M * comm.world, ptr(dz), stream())
        return dz, loc[:N], loc[N:]
    dgamma = torch.empty(N, dtype=torch.float32, device=dh.device)
    dbeta = torch.empty(N, dtype=torch.float32, device=dh.device)
    call("rs_bn_bwd", ptr(dh), ptr(h), ptr(xhat), M, N, int(act), ptr(gamma), ptr(var), float(eps), ptr(dgamma),
         ptr(dbeta), ptr(dz), stream())
    return dz, dgamma, dbeta


def ln_fwd(z, gamma, beta, act: int, eps=1e-5):
    M, N = z.shape
    h = torch.empty_like(z)
    xhat = torch.empty_like(z)
    rstd = torch.empty(M, dtype=torch.float32, device=z.device)
    call("rs_ln_fwd", ptr(z), M, N, ptr(gamma), ptr(beta), float(eps), int(act), ptr(xhat), ptr(rstd), ptr(h), stream())
    return h, xhat, rstd


def ln_bwd(dh, h, xhat, rstd, gamma, act: int):
    M, N = dh.shape
    dz = torch.empty_like(dh)
    dy = torch.empty_like(dh)
    dyx = torch.empty_like(dh)
    call("rs_ln_bwd", ptr(dh), ptr(h), ptr(xhat), ptr(rstd), ptr(gamma), M, N, int(act), ptr(dz), ptr(dy), ptr(dyx),
         stream())
    return dz, colsum(dyx), colsum(dy)


def dropout_apply(x: torch.Tensor, p: float, seed: int, salt: int, call_no: int = 0,
                  call_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x * mask / (1 - p) with the mask of (seed, salt, call_no + *call_dev); applied to a gradient with the same numbers
    it is the backward pass of the dropout that produced the activation."""
    _chk(x, torch.float32, "x")
    if call_dev is not None:
        _chk(call_dev, torch.int64, "call_dev", (1,))
    y = out if out is not None else torch.empty_like(x)
    call("rs_dropout_apply", ptr(x), ptr(y), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(salt) & 0xFFFFFFFF,
         int(call_no) & 0xFFFFFFFFFFFFFFFF, ptr(call_dev), stream())
    return y


def add(a, b):
    y = torch.empty_like(a)
    call("rs_add_f32", ptr(a), ptr(b), ptr(y), a.numel(), stream())
    return y


# ---------------------------------------------------------------------------------------------
# interactions on gathered embeddings
# ---------------------------------------------------------------------------------------------
def fm_fwd(emb: torch.Tensor, want_sum=True, want_bi=False, want_fm=True):
    _chk(emb, torch.float32, "emb")
    B, Fn, d = emb.shape
    dev = emb.device
    out = {}
    if want_sum:
        out["sum"] = torch.empty((B, d), dtype=torch.float32, device=dev)
    if want_bi:
        out["bi"] = torch.empty((B, d), dtype=torch.float32, device=dev)
    if want_fm:
        out["fm"] = torch.empty((B,), dtype=torch.float32, device=dev)
    call("rs_fm_fwd", ptr(emb), B, Fn, d, ptr(out.get("sum")), ptr(out.get("bi")), ptr(out.get("fm")), stream())
    return out


def fm_bwd(emb, S, dfm=None, dbi=None):
    B, Fn, d = emb.shape
    _chk(S, torch.float32, "S", (B, d))
    _chk(dfm, torch.float32, "dfm", (B,))
    _chk(dbi, torch.float32, "dbi", (B, d))
    demb = torch.empty_like(emb)
    call("rs_fm_bwd", ptr(emb), ptr(S), ptr(dfm), ptr(dbi), B, Fn, d, ptr(demb), stream())
    return demb


def interact_partial_rows(B: int) -> int:
    return _lib.lib().rs_interact_partial_rows(int(B))


def pnn_inner_fwd(emb: torch.Tensor) -> torch.Tensor:
    _chk(emb, torch.float32, "emb")
    B, Fn, d = emb.shape
    out = torch.empty((B, Fn * (Fn - 1) // 2), dtype=torch.float32, device=emb.device)
    call("rs_pnn_inner_fwd", ptr(emb), B, Fn, d, ptr(out), stream())
    return out


def pnn_inner_bwd(emb: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    B, Fn, d = emb.shape
    _chk(dout, torch.float32, "dout", (B, Fn * (Fn - 1) // 2))
    demb = torch.empty_like(emb)
    call("rs_pnn_inner_bwd", ptr(emb), ptr(dout), B, Fn, d, ptr(demb), stream())
    return demb


def pnn_outer_fwd(emb: torch.Tensor) -> torch.Tensor:
    _chk(emb, torch.float32, "emb")
    B, Fn, d = emb.shape
    out = torch.empty((B, Fn * (Fn - 1) // 2 * d * d), dtype=torch.float32, device=emb.device)
    call("rs_pnn_outer_fwd", ptr(emb), B, Fn, d, ptr(out), stream())
    return out


def pnn_outer_bwd(emb: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    B, Fn, d = emb.shape
    _chk(dout, torch.float32, "dout", (B, Fn * (Fn - 1) // 2 * d * d))
    demb = torch.empty_like(emb)
    call("rs_pnn_outer_bwd", ptr(emb), ptr(dout), B, Fn, d, ptr(demb), stream())
    return demb


def pnn_inner_cat_ok(Fn: int, d: int) -> bool:
    return bool(_lib.lib().rs_pnn_inner_cat_ok(int(Fn), int(d)))


def pnn_inner_cat_fwd(emb: torch.Tensor) -> torch.Tensor:
    """[B, ceil4(F*d + P)] rows = [emb row | inner products of all field pairs | zeros]: the (padded) input of PNN's
    first MLP layer with pnn_type inner, in one launch."""
    B, Fn, d = emb.shape
    _chk(emb, torch.float32, "emb")
    K = Fn * d + Fn * (Fn - 1) // 2
    ld = (K + 3) // 4 * 4
    out = torch.empty((B, ld), dtype=torch.float32, device=emb.device)
    call("rs_pnn_inner_cat_fwd", ptr(emb), B, Fn, d, ptr(out), ld, stream())
    return out


def pnn_inner_cat_bwd(emb: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    """demb of pnn_inner_cat_fwd from the gradient of its rows."""
    B, Fn, d = emb.shape
    ld = _chk_rows(dout, "dout")
    if dout.shape[0] != B or dout.shape[1] < Fn * d + Fn * (Fn - 1) // 2:
        raise RSError("pnn_inner_cat_bwd: dout does not hold the concatenated rows")
    demb = torch.empty_like(emb)
    call("rs_pnn_inner_cat_bwd", ptr(emb), ptr(dout), ld, B, Fn, d, ptr(demb), stream())
    return demb


def pair_start(i: int, F: int) -> int:
    """Index of pair (i, i+1) in the first-field-major numbering of the F(F-1)/2 pairs (PNN.py:83-85)."""
    return i * (2 * F - i - 1) // 2


def pnn_outer_fwd_range(emb: torch.Tensor, i_lo: int, i_hi: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The outer products of the pairs (i, j), i in [i_lo, i_hi): [B, n_pairs * d * d] (out: a buffer with contiguous
    rows of at least that length)."""
    _chk(emb, torch.float32, "emb")
    B, Fn, d = emb.shape
    width = (pair_start(i_hi, Fn) - pair_start(i_lo, Fn)) * d * d
    if out is None:
        out = torch.empty((B, width), dtype=torch.float32, device=emb.device)
    ld = _chk_rows(out, "out", (B, width))
    call("rs_pnn_outer_fwd_range", ptr(emb), B, Fn, d, int(i_lo), int(i_hi), ptr(out), ld, stream())
    return out


def pnn_outer_bwd_range(emb: torch.Tensor, dout: torch.Tensor, i_lo: int, i_hi: int, demb: torch.Tensor,
                        accumulate: bool) -> torch.Tensor:
    """demb (+)= the gradient the products of pairs (i, j), i in [i_lo, i_hi), send back to the embeddings."""
    _chk(emb, torch.float32, "emb")
    B, Fn, d = emb.shape
    width = (pair_start(i_hi, Fn) - pair_start(i_lo, Fn)) * d * d
    ld = _chk_rows(dout, "dout", (B, width))
    _chk(demb, torch.float32, "demb", (B, Fn, d))
    call("rs_pnn_outer_bwd_range", ptr(emb), ptr(dout), ld, B, Fn, d, int(i_lo), int(i_hi), int(bool(accumulate)),
         ptr(demb), stream())
    return demb


def pnn_outer_mlp_fwd(emb, x_other, W, bias, block_bytes: int = 0, ws: Optional[Workspace] = None) -> torch.Tensor:
    """z = [x_other | outer products of all field pairs of emb] W^T + bias without the matrix of products (a block of
    pairs at a time inside the library; block_bytes: size of a block of products, 0 = the library's default).
    emb [B,F,d]; x_other [B,K0] (rows contiguous) or None; W [N, K0 + P*d*d]."""
    B, Fn, d = emb.shape
    N = W.shape[0]
    K0 = 0 if x_other is None else x_other.shape[1]
    _chk(emb, torch.float32, "emb")
    _chk(W, torch.float32, "W", (N, K0 + Fn * (Fn - 1) // 2 * d * d))
    _chk(bias, torch.float32, "bias", (N,))
    ldx = _chk_rows(x_other, "x_other", (B, K0)) if x_other is not None else 0
    n = _lib.lib().rs_pnn_outer_mlp_workspace_floats(B, Fn, d, K0, N, 0, int(block_bytes))
    if n == 0:
        raise RSError("pnn_outer_mlp: unsupported shape")
    wbuf = (ws or default_workspace(emb.device)).get("pnn_outer_mlp", n)
    z = torch.empty((B, N), dtype=torch.float32, device=emb.device)
    call("rs_pnn_outer_mlp_fwd", ptr(emb), ptr(x_other), ldx, B, Fn, d, K0, ptr(W), ptr(bias), N, ptr(z), ptr(wbuf), n,
         int(block_bytes), stream())
    return z


def pnn_outer_mlp_bwd(emb, x_other, dz, W, dW_out=None, db_out=None, want_dx_other=True, block_bytes: int = 0,
                      ws: Optional[Workspace] = None):
    """-> (dW [N, K0 + P*d*d], db [N], demb [B,F,d], dx_other [B,K0] or None) of pnn_outer_mlp_fwd."""
    B, Fn, d = emb.shape
    N = W.shape[0]
    K0 = 0 if x_other is None else x_other.shape[1]
    _chk(dz, torch.float32, "dz", (B, N))
    ldx = _chk_rows(x_other, "x_other", (B, K0)) if x_other is not None else 0
    n = _lib.lib().rs_pnn_outer_mlp_workspace_floats(B, Fn, d, K0, N, 1, int(block_bytes))
    if n == 0:
        raise RSError("pnn_outer_mlp: unsupported shape")
    wbuf = (ws or default_workspace(emb.device)).get("pnn_outer_mlp", n)
    dW = dW_out if dW_out is not None else torch.empty_like(W)
    db = db_out if db_out is not None else torch.empty((N,), dtype=torch.float32, device=emb.device)
    _chk(dW, torch.float32, "dW", tuple(W.shape))
    _chk(db, torch.float32, "db", (N,))
    demb = torch.empty_like(emb)
    dxo = torch.empty((B, K0), dtype=torch.float32, device=emb.device) if (K0 and want_dx_other) else None
    call("rs_pnn_outer_mlp_bwd", ptr(emb), ptr(x_other), ldx, ptr(dz), B, Fn, d, K0, ptr(W), N, ptr(dW), ptr(db), ptr(demb),
         ptr(dxo), ptr(wbuf), n, int(block_bytes), stream())
    return dW, db, demb, dxo


def dcn_cross_fwd(x0: torch.Tensor, w: torch.Tensor, bias: torch.Tensor):
    """x0 [B,D]; w, bias [L,D] -> (out [B,D], S [B,L])."""
    _chk(x0, torch.float32, "x0")
    B, D = x0.shape
    L = w.shape[0]
    _chk(w, torch.float32, "w", (L, D))
    _chk(bias, torch.float32, "bias", (L, D))
    out = torch.empty_like(x0)
    S = torch.empty((B, L), dtype=torch.float32, device=x0.device)
    call("rs_dcn_cross_fwd", ptr(x0), ptr(w), ptr(bias), B, D, L, ptr(out), ptr(S), stream())
    return out, S


def dcn_cross_bwd(x0, S, dout, w, bias, g_out=None, parts: "ColsumParts" = None, dx_add=None):
    """-> (dx0 [B,D], dW [L,D], dB [L,D]); g_out: optional [2*L*D] destination of (dW, dB) laid end to end; dx_add [B,D]:
    added into dx0 (the gradient x0 receives from elsewhere)."""
    B, D = x0.shape
    L = w.shape[0]
    _chk(dout, torch.float32, "dout", (B, D))
    _chk(g_out, torch.float32, "g_out", (2 * L * D,))
    _chk(dx_add, torch.float32, "dx_add", (B, D))
    rows = interact_partial_rows(B)
    dx0 = torch.empty_like(x0)
    part = torch.empty((rows, 2 * L * D), dtype=torch.float32, device=x0.device)
    if dx_add is not None:
        call("rs_dcn_cross_bwd_add", ptr(x0), ptr(S), ptr(dout), ptr(w), ptr(bias), B, D, L, ptr(dx_add), ptr(dx0), ptr(part),
             rows, stream())
    else:
        call("rs_dcn_cross_bwd", ptr(x0), ptr(S), ptr(dout), ptr(w), ptr(bias), B, D, L, ptr(dx0), ptr(part), rows, stream())
    g = colsum(part, out=g_out, parts=parts if g_out is not None else None)     # with parts: valid after its finish()
    return dx0, g[:L * D].view(L, D), g[L * D:].view(L, D)


def nfm_bi_fwd(E: torch.Tensor, V: torch.Tensor):
    """E [B,D], V [D,K] -> (out [B,K], s [B,K])."""
    _chk(E, torch.float32, "E")
    B, D = E.shape
    K = V.shape[1]
    _chk(V, torch.float32, "V", (D, K))
    out = torch.empty((B, K), dtype=torch.float32, device=E.device)
    s = torch.empty_like(out)
    call("rs_nfm_bi_fwd", ptr(E), ptr(V), B, D, K, ptr(out), ptr(s), stream())
    return out, s


def nfm_bi_bwd(E, V, s, dout):
    B, D = E.shape
    K = V.shape[1]
    _chk(dout, torch.float32, "dout", (B, K))
    rows = interact_partial_rows(B)
    dE = torch.empty_like(E)
    part = torch.empty((rows, D * K), dtype=torch.float32, device=E.device)
    call("rs_nfm_bi_bwd", ptr(E), ptr(V), ptr(s), ptr(dout), B, D, K, ptr(dE), ptr(part), rows, stream())
    return dE, colsum(part).view(D, K)


def afm_fwd(emb, W, bA, h, pw, pb) -> torch.Tensor:
    """emb [B,F,d]; W [A,d], bA [A], h [A] (or [1,A]), pw [d] (or [1,d]), pb [1] -> out [B]."""
    _chk(emb, torch.float32, "emb")
    B, Fn, d = emb.shape
    A = W.shape[0]
    _chk(W, torch.float32, "W", (A, d))
    for t, n, k in ((bA, "bA", A), (h, "h", A), (pw, "pw", d), (pb, "pb", 1)):
        _chk(t, torch.float32, n)
        if t.numel() != k:
            raise RSError(f"afm: {n} must have {k} elements")
    out = torch.empty((B,), dtype=torch.float32, device=emb.device)
    call("rs_afm_fwd", ptr(emb), B, Fn, d, A, ptr(W), ptr(bA), ptr(h), ptr(pw), ptr(pb), ptr(out), stream())
    return out


def afm_bwd(emb, dout, W, bA, h, pw, g_out=None, parts: "ColsumParts" = None):
    """-> (demb, dW [A,d], dbA [A], dh [A], dpw [d], dpb [1]); g_out: optional [A*d + 2A + d + 1] destination of the five
    parameter gradients laid end to end in that order."""
    B, Fn, d = emb.shape
    A = W.shape[0]
    _chk(dout, torch.float32, "dout", (B,))
    rows = interact_partial_rows(B)
    np_ = A * d + 2 * A + d + 1
    _chk(g_out, torch.float32, "g_out", (np_,))
    demb = torch.empty_like(emb)
    part = torch.empty((rows, np_), dtype=torch.float32, device=emb.device)
    call("rs_afm_bwd", ptr(emb), ptr(dout), B, Fn, d, A, ptr(W), ptr(bA), ptr(h), ptr(pw), ptr(demb), ptr(part), rows,
         stream())
    g = colsum(part, out=g_out, parts=parts if g_out is not None else None)     # with parts: valid after its finish()
    o = A * d
    return demb, g[:o].view(A, d), g[o:o + A], g[o + A:o + 2 * A], g[o + 2 * A:o + 2 * A + d], g[o + 2 * A + d:]


def ffm_fwd(idx: torch.Tensor, layout: FieldLayout, T: torch.Tensor) -> torch.Tensor:
    """idx [F,B] int64; T [sum V_f, F*d] -> out [B]."""
    Fn, B = idx.shape
    _chk(idx, torch.int64, "idx", (layout.num_fields, B))
    _chk(T, torch.float32, "T")
    if T.shape[0] != layout.total_rows or T.shape[1] % Fn:
        raise RSError("ffm: table must be [sum vocab, F*d]")
    d = T.shape[1] // Fn
    out = torch.empty((B,), dtype=torch.float32, device=idx.device)
    call("rs_ffm_fwd", ptr(idx), ptr(layout.row_offset_t), ptr(layout.vocab_t), ptr(T), B, Fn, d, ptr(out),
         ptr(layout.err_flag), stream())
    return out


def ffm_bwd(plan: "SparsePlan", idx: torch.Tensor, T: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    """Dense gradient of T; `plan` is a SparsePlan of the same layout (it is re-sorted here)."""
    Fn, B = idx.shape
    d = T.shape[1] // Fn
    _chk(dout, torch.float32, "dout", (B,))
    plan.sort_columnar(idx)
    dT = torch.zeros_like(T)
    call("rs_ffm_bwd", ptr(plan.plan), ptr(idx), ptr(plan.layout.row_offset_t), ptr(plan.layout.vocab_t), ptr(T), ptr(dout),
         B, Fn, plan.cap, d, ptr(dT), stream())
    return dT


class WidePlan(SparsePlan):
    """Sort plan + reduction buffers for gradient rows of W floats (W % 4 == 0, W <= 1024): the wide-row form of
    SparsePlan.reduce / update used by the row-sharded exchange of field-aware FFM rows."""

    def __init__(self, layout: FieldLayout, cap: int, W: int, device=None):
        super().__init__(layout, cap, 1, with_lr=False, device=device)
        if W % 4 or W > 1024:
            raise RSError(f"wide rows: width {W} not in 4..1024 step 4")
        if self.CHUNK > 64:
            raise RSError("wide rows: plan window exceeds 64 entries")
        self.width = int(W)
        dev = device or layout.device
        self.rowgrad_wide = torch.empty((self.F, self.cap, self.width), dtype=torch.float32, device=dev)
        self.chunk_part_wide = torch.empty((self.F, self.cap, self.width), dtype=torch.float32, device=dev)
        self.n_partials_wide = _lib.lib().rs_sparse_bwd_wide_num_partials(self.F, self.cap)

    def reduce_wide(self, norm_partials: torch.Tensor, gout: torch.Tensor):
        _chk(gout, torch.float32, "gout")
        if gout.shape[-1] != self.width:
            raise RSError("wide rows: gradient rows have the wrong width")
        if norm_partials.numel() < self.n_partials_wide:
            raise RSError("wide rows: norm_partials too small")
        call("rs_sparse_bwd_reduce_wide", ptr(self.plan), self.F, self.cap, self.width, ptr(gout), ptr(self.rowgrad_wide),
             ptr(self.chunk_part_wide), ptr(norm_partials), stream())

    def update_wide(self, opt: str, hyper, scal, T, m=None, v=None, rs=None):
        call("rs_sparse_update", ptr(self.plan), self.F, self.cap, self.width, OPT_CODES[opt], ptr(self.rowgrad_wide),
             ptr(None), ptr(T), ptr(m), ptr(v), ptr(None), ptr(None), ptr(None), ptr(hyper), ptr(scal),
             *self._stamps(rs, None), stream())


def ffm_rows_fwd(G: torch.Tensor, d: int) -> torch.Tensor:
    """G [B, F, F*d] gathered field-aware rows -> out [B]."""
    _chk(G, torch.float32, "G")
    B, Fn, W = G.shape
    if W != Fn * d:
        raise RSError("ffm_rows_fwd: rows must be F*d wide")
    out = torch.empty((B,), dtype=torch.float32, device=G.device)
    call("rs_ffm_rows_fwd", ptr(G), B, Fn, d, ptr(out), stream())
    return out


def ffm_rows_bwd(G: torch.Tensor, dout: torch.Tensor, d: int) -> torch.Tensor:
    """Gradient with respect to the gathered rows, [B, F, F*d]."""
    B, Fn, W = G.shape
    _chk(dout, torch.float32, "dout", (B,))
    dG = torch.empty_like(G)
    call("rs_ffm_rows_bwd", ptr(G), ptr(dout), B, Fn, d, ptr(dG), stream())
    return dG


class FFMPlan(SparsePlan):
    """Sort plan of a batch plus the compact buffer of field-aware row gradients (rows of F*d floats)."""

    def __init__(self, layout: FieldLayout, cap: int, d: int, with_lr: bool = False, device=None):
        super().__init__(layout, cap, 1, with_lr=with_lr, device=device)
        if d % 4:
            raise RSError(f"FFM touched-rows update: embedding_dim {d} is not a multiple of 4")
        self.ffm_d = int(d)
        self.width = self.F * self.ffm_d
        if self.width > 1024:
            raise RSError(f"FFM touched-rows update: rows of {self.width} floats exceed 1024")
        dev = device or layout.device
        if self.CHUNK > 64:
            raise RSError("FFM touched-rows update: plan window exceeds 64 entries")
        self.rowgrad_wide = torch.empty((self.F, self.cap, self.width), dtype=torch.float32, device=dev)
        self.chunk_part_wide = torch.empty((self.F, self.cap, self.width), dtype=torch.float32, device=dev)
        self.n_partials_wide = _lib.lib().rs_ffm_bwd_num_partials(self.F, self.cap)

    def rows(self, idx: torch.Tensor, T: torch.Tensor, dout: torch.Tensor, norm_partials: torch.Tensor,
             sorted_already: bool = False):
        """Sorts idx (unless the plan already holds its sort), then fills the compact row gradients and
        norm_partials[:n_partials_wide]."""
        Fn, B = idx.shape
        _chk(T, torch.float32, "T", (self.layout.total_rows, self.width))
        _chk(dout, torch.float32, "dout", (B,))
        if norm_partials.numel() < self.n_partials_wide:
            raise RSError("FFM backward: norm_partials too small")
        if not sorted_already:
            self.sort_columnar(idx)
        call("rs_ffm_bwd_rows", ptr(self.plan), ptr(idx), ptr(self.layout.row_offset_t), ptr(self.layout.vocab_t), ptr(T),
             ptr(dout), B, self.F, self.cap, self.ffm_d, ptr(self.rowgrad_wide), ptr(self.chunk_part_wide), ptr(norm_partials),
             stream())

    def update_wide(self, opt: str, hyper, scal, T, m=None, v=None, rs=None):
        call("rs_sparse_update", ptr(self.plan), self.F, self.cap, self.width, OPT_CODES[opt], ptr(self.rowgrad_wide),
             ptr(None), ptr(T), ptr(m), ptr(v), ptr(None), ptr(None), ptr(None), ptr(hyper), ptr(scal),
             *self._stamps(rs, None), stream())

    def scatter_wide(self, dT: torch.Tensor):
        call("rs_sparse_scatter_dense", ptr(self.plan), self.F, self.cap, self.width, ptr(self.rowgrad_wide), ptr(None),
             ptr(dT), ptr(None), stream())


def axpy(a: float, x: torch.Tensor, y: torch.Tensor):
    """y += a * x (flat)."""
    _chk(x, torch.float32, "x")
    _chk(y, torch.float32, "y", x.shape)
    call("rs_axpy_f32", float(a), ptr(x), ptr(y), x.numel(), stream())
