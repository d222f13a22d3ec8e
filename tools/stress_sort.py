"""Randomised comparison of the index-sort forms (one workgroup per field / tiled / merge of sorted lists): the plans
must be identical bit for bit.  Not part of the test suite; run on the GPU box after touching the sort."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendsystem_amd import _lib, ops  # noqa: E402


def main(cases=150):
    dev = "cuda:0"
    h = _lib.lib()
    g = torch.Generator().manual_seed(1234)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for c in range(cases):
        F = ri(1, 12)
        vocabs = [ri(2, 10 ** ri(1, 7)) for _ in range(F)]
        layout = ops.FieldLayout(vocabs, dev)
        if c % 2 == 0:
            B = ri(1, 16384)
            skew = ri(1, 5)
            idx = torch.stack([((torch.rand(B, generator=g) ** skew) * v).long().clamp(0, v - 1) for v in vocabs]).to(dev)
            cap = min(16384, B + ri(0, 100))
            plans = []
            for on in (0, 1):
                h.rs_sort_two_level(on)
                p = ops.SparsePlan(layout, cap=cap, d=[4, 8, 16, 1][c % 4], with_lr=False)
                p.sort_columnar(idx)
                plans.append(p)
            h.rs_sort_two_level(1)
            used = F * (6 * cap + 6)
            assert torch.equal(plans[0].plan[:used], plans[1].plan[:used]), f"case {c}: tiled sort differs (B={B}, F={F})"
        else:
            N = ri(1, 16)
            cap_b = ri(1, 16384 // N)
            keys = torch.zeros(N * F, cap_b + 1, dtype=torch.int64)
            cnt = torch.zeros(N * F, dtype=torch.int32)
            for s in range(N):
                for f, v in enumerate(vocabs):
                    want = ri(0, cap_b)
                    ks = torch.unique(torch.randint(1, max(v, 2), (want,), generator=g)) if want and v > 1 else torch.zeros(0, dtype=torch.int64)
                    ks = ks[ks < v]
                    keys[s * F + f, 1:1 + len(ks)] = ks
                    cnt[s * F + f] = len(ks)
            sf = torch.arange(N * F, dtype=torch.int32)
            sub_off, pay_base = (sf * (cap_b + 1) + 1).to(dev), (sf * cap_b).to(dev)
            cap = N * cap_b
            ps = ops.SparsePlan(layout, cap=cap, d=16, with_lr=False)
            pm = ops.SparsePlan(layout, cap=cap, d=16, with_lr=False)
            ps.sort_lists(keys.to(dev).view(-1), sub_off, cnt.to(dev), pay_base, N, cap)
            pm.merge_lists(keys.to(dev).view(-1), sub_off, cnt.to(dev), pay_base, N, cap)
            used = F * (6 * cap + 6)
            assert torch.equal(ps.plan[:used], pm.plan[:used]), f"case {c}: merge differs (N={N}, cap_b={cap_b}, F={F})"
        if c % 10 == 0:
            print("case", c, "ok", flush=True)
    print("all", cases, "cases identical")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 150)
