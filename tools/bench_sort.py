"""Time the per-field index sort (one-pass against two-level) on Criteo_x1-shaped batches."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendsystem_amd import _lib, data, ops  # noqa: E402


def main():
    dev = "cuda:0"
    vocabs = data.criteo_x1_vocab_sizes()
    layout = ops.FieldLayout(vocabs, dev)
    h = _lib.lib()
    g = torch.Generator().manual_seed(0)
    for B in (512, 1024, 2048, 4096, 8192, 16384):
        idx = torch.stack([((torch.rand(B, generator=g) ** 3) * v).long().clamp(0, v - 1) for v in vocabs]).to(dev)
        plan = ops.SparsePlan(layout, cap=B, d=16, with_lr=False)
        for on in (0, 1):
            h.rs_sort_two_level(on)
            for _ in range(5):
                plan.sort_columnar(idx)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                plan.sort_columnar(idx)
            e1.record()
            torch.cuda.synchronize()
            print(f"B={B} two_level={on} sort {e0.elapsed_time(e1) / 50 * 1e3:.1f} us", flush=True)
        h.rs_sort_two_level(1)

    # owner side of the row-sharded exchange: N ascending duplicate-free lists per field, sort against merge
    for N in (2, 4, 8):
        B = 4096
        cap_b = (3 * B) // (2 * N) + 64
        F = len(vocabs)
        keys = torch.zeros(N * F, cap_b + 1, dtype=torch.int64)
        cnt = torch.zeros(N * F, dtype=torch.int32)
        for s_ in range(N):
            for f, v in enumerate(vocabs):
                lv = max(2, v // N)
                k = torch.unique(((torch.rand(B, generator=g) ** 3) * v).long().clamp(1, v - 1))
                k = torch.unique(k[k % N == 0] // N + 1).clamp(max=lv - 1)[:cap_b]
                k = torch.unique(k)
                keys[s_ * F + f, 1:1 + len(k)] = k
                cnt[s_ * F + f] = len(k)
        sf = torch.arange(N * F, dtype=torch.int32)
        sub_off, pay_base = (sf * (cap_b + 1) + 1).to(dev), (sf * cap_b).to(dev)
        lay = ops.FieldLayout([max(2, v // N) + 1 for v in vocabs], dev)
        plan = ops.SparsePlan(lay, cap=N * cap_b, d=16, with_lr=False)
        kd, cd = keys.to(dev).view(-1), cnt.to(dev)
        for how in ("sort_lists", "merge_lists"):
            fn = getattr(plan, how)
            for _ in range(5):
                fn(kd, sub_off, cd, pay_base, N, N * cap_b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn(kd, sub_off, cd, pay_base, N, N * cap_b)
            e1.record()
            torch.cuda.synchronize()
            print(f"owner side N={N} cap={N * cap_b} entries~{int(cnt.view(N, F).sum(0).float().mean())}/field "
                  f"{how} {e0.elapsed_time(e1) / 50 * 1e3:.1f} us", flush=True)


if __name__ == "__main__":
    main()
