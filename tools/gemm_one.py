"""Run one GEMM shape repeatedly (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommendsystem_amd import ops
amc, bnc, M, N, K, tile, sk, iters = [int(x) for x in sys.argv[1:9]]
dev = "cuda:0"
a = torch.randn((K, M) if amc else (M, K), device=dev)
b = torch.randn((K, N) if bnc else (N, K), device=dev)
for _ in range(iters):
    ops.gemm(a, b, a_mc=bool(amc), b_nc=bool(bnc), tile=tile, splitk=sk)
torch.cuda.synchronize()
