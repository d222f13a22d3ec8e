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


if __name__ == "__main__":
    main()
