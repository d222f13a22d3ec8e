"""Per-launch kernel timing with HIP events recorded on the launch stream.

``KernelTimer`` wraps the single funnel through which every C-ABI entry point is called
(``_lib.call``) and brackets the selected entry points with ``torch.cuda.Event`` pairs recorded on
torch's current stream -- the stream the kernels are launched on.  Durations are read after a
device synchronisation, so nothing is synchronised inside the measured region.
"""
from __future__ import annotations

from collections import defaultdict

import torch

from . import _lib, ops


def _gemm_key(args):
    a_mc, b_nc, M, N, K = args[0], args[1], args[2], args[3], args[4]
    return ("rs_gemm_f32", f"{'T' if a_mc else 'N'}{'N' if b_nc else 'T'} M={M} N={N} K={K}"), 2.0 * M * N * K, "flop"


def _gather_key(args):
    B, F, d = args[11], args[12], args[13]
    has_table, has_t1 = args[3] is not None, args[4] is not None
    emb_out, sum_out, lr_out, fm_out = args[6], args[7], args[9], args[10]
    byts = B * F * 8
    if has_table:
        byts += B * F * d * 4
    if has_t1:
        byts += B * F * 4
    if emb_out is not None:
        byts += B * F * d * 4
    if sum_out is not None:
        byts += B * d * 4
    byts += (4 * B if lr_out is not None else 0) + (4 * B if fm_out is not None else 0)
    return ("rs_gather_lr_fm_fwd", f"B={B} F={F} d={d}"), float(byts), "byte"


def _group_key(args):
    # (descriptor array, count, stages, stream): the products of one grouped launch and their summed work
    descs, count = args[0], args[1]
    parts, work = [], 0.0
    for i in range(count):
        d = descs[i]
        parts.append(f"{'T' if d.a_mc else 'N'}{'N' if d.b_nc else 'T'}:M={d.M},N={d.N},K={d.K}")
        work += 2.0 * d.M * d.N * d.K
    return ("rs_gemm_f32_group", "GRP " + " ".join(parts)), work, "flop"


KEYERS = {"rs_gemm_f32": _gemm_key, "rs_gemm_f32_group": _group_key, "rs_gather_lr_fm_fwd": _gather_key}


class KernelTimer:
    def __init__(self, names=None):
        """names: entry points to time (None: every entry point that takes a stream)."""
        self.names = set(names) if names is not None else set(_lib._SIGS)
        self.records = []          # (key, work, unit, start_event, end_event)
        self.overhead_us = 0.0     # what an event pair adds to the duration of the kernel between its records

    def calibrate(self, pairs: int = 48):
        """What an event pair adds to the duration of the kernel between its two records (subtracted from every
        record): pairs around spin kernels of two known lengths -- the duration at length zero of the line through the
        two medians.  (A pair with nothing in between reads longer than that: the second record then waits for the
        first one's own completion, which a kernel in between hides.)"""
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(4000000); e1.record(); torch.cuda.synchronize()
        cyc_per_us = 4000000 / max(e0.elapsed_time(e1) * 1e3, 1.0)
        med = []
        for us in (10.0, 40.0):
            ev = []
            torch.cuda._sleep(int(2000 * cyc_per_us))      # let the host run ahead of the GPU, as in the measured pass
            for _ in range(pairs):
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                torch.cuda._sleep(int(us * cyc_per_us))
                e.record()
                ev.append((s, e))
            torch.cuda.synchronize()
            d = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
            med.append(d[len(d) // 2])
        slope = (med[1] - med[0]) / 30.0
        self.overhead_us = max(med[0] - 10.0 * slope, 0.0)
        return self.overhead_us

    def __enter__(self):
        self._orig = _lib.call
        timer = self

        def timed_call(name, *args):
            if name not in timer.names:
                return timer._orig(name, *args)
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = timer._orig(name, *args)
            e.record()
            if name in KEYERS:
                key, work, unit = KEYERS[name](args)
            else:
                key, work, unit = (name, ""), 0.0, ""
            timer.records.append((key, work, unit, s, e))
            return r

        _lib.call = timed_call
        ops.call = timed_call
        return self

    def __exit__(self, *exc):
        _lib.call = self._orig
        ops.call = self._orig
        return False

    def summary(self):
        """{(entry point, shape): dict(n, avg_us, total_us, raw_avg_us, work_per_launch, unit)} after a device sync;
        raw_* are the event durations as read, the others have the pair's own overhead (calibrate) subtracted."""
        torch.cuda.synchronize()
        agg = defaultdict(lambda: {"n": 0, "total_us": 0.0, "raw_total_us": 0.0, "work": 0.0, "unit": ""})
        for key, work, unit, s, e in self.records:
            a = agg[key]
            a["n"] += 1
            a["total_us"] += max(s.elapsed_time(e) * 1e3 - self.overhead_us, 0.0)
            a["raw_total_us"] += s.elapsed_time(e) * 1e3
            a["work"] = work
            a["unit"] = unit
        for a in agg.values():
            a["avg_us"] = a["total_us"] / max(a["n"], 1)
            a["raw_avg_us"] = a["raw_total_us"] / max(a["n"], 1)
        return dict(agg)
