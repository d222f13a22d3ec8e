"""Does co-residency help?  Same K, tile count = 1, 2, 3, 4, 6 per CU (64x64 tiles, 256 CUs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommendsystem_amd import ops
dev = "cuda:0"
def bench(M, N, K, reps=30):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
    for _ in range(3): ops.gemm(a, b, tile=3, splitk=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(a, b, tile=3, splitk=1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for K in (624, 2048):
    for per_cu in (1, 2, 3, 4, 6, 8):
        M = 1024 * per_cu
        us = bench(M, 1024, K)
        print(f"K={K:5d} tiles/CU={per_cu}  {us:7.1f} us  {2.0 * M * 1024 * K / us / 1e6:6.1f} TFLOP/s", flush=True)
