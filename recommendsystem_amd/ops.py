"""Tensor-level wrappers over the C-ABI (no autograd here; see functional.py).

Every function takes CUDA (ROCm) tensors, validates shapes/dtypes on the host so the kernels
never see an operand they were not written for, allocates outputs with torch, and launches
on torch's current stream.
"""
from __future__ import annotations

from typing import Optional

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
                     want_emb=True, want_sum=False, want_bi=False, want_lr=False, want_fm=False):
    """Fused lookup; returns dict with the requested outputs (emb, sum, bi, lr, fm)."""
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
    call("rs_gather_lr_fm_fwd", ptr(idx), ptr(layout.row_offset_t), ptr(layout.vocab_t), ptr(table), ptr(table1),
         ptr(bias), ptr(out.get("emb")), ptr(out.get("sum")), ptr(out.get("bi")), ptr(out.get("lr")),
         ptr(out.get("fm")), B, Fn, d, ptr(layout.err_flag), stream())
    return out
