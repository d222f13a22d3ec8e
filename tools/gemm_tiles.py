"""Time every tile / split choice on the nine GEMM shapes of the DeepFM step (back-to-back launches, one event pair)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommendsystem_amd import ops

dev = "cuda:0"
SHAPES = [("NT", 4096, 1024, 624), ("NT", 4096, 512, 1024), ("NT", 4096, 256, 512),
          ("NN", 4096, 624, 1024), ("NN", 4096, 1024, 512), ("NN", 4096, 512, 256),
          ("TN", 1024, 624, 4096), ("TN", 512, 1024, 4096), ("TN", 256, 512, 4096)]

def bench(lay, M, N, K, tile, sk, reps=30):
    amc, bnc = lay[0] == "T", lay[1] == "N"
    a = torch.randn((K, M) if amc else (M, K), device=dev)
    b = torch.randn((K, N) if bnc else (N, K), device=dev)
    for _ in range(3):
        ops.gemm(a, b, a_mc=amc, b_nc=bnc, tile=tile, splitk=sk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(a, b, a_mc=amc, b_nc=bnc, tile=tile, splitk=sk)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for lay, M, N, K in SHAPES:
    row = []
    sks = [1] if lay != "TN" else [2, 4, 8, 16]
    if lay != "TN" and K >= 512 and M * N <= 4096 * 512:
        sks = [1, 2]
    for tile in (0, 1, 2, 3, 5):
        for sk in sks:
            us = bench(lay, M, N, K, tile, sk)
            row.append(f"t{tile}/k{sk}:{us:6.1f}")
    print(f"{lay} {M:5d}x{N:5d}x{K:5d}  ideal {2.0 * M * N * K / 157.3e6:5.1f}  " + "  ".join(row), flush=True)
