"""Study aid: fp32 GEMM rates of the vendor BLAS (through torch.mm) on the step's nine GEMM shapes, beside
rs_gemm_f32 on the same operands.  Not used by the product path.

    python tools/blas_compare.py
"""
import json
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendsystem_amd import ops  # noqa: E402

B = 4096
SHAPES = []      # (name, a_mc, b_nc, M, N, K)
dims = [624, 1024, 512, 256]
for i in range(3):
    fin, fout = dims[i], dims[i + 1]
    SHAPES.append((f"fwd{i+1}", False, False, B, fout, fin))      # X[B,in] @ W[out,in]^T   (NT)
    SHAPES.append((f"dX{i+1}", False, True, B, fin, fout))        # dZ[B,out] @ W[out,in]    (NN)
    SHAPES.append((f"dW{i+1}", True, True, fout, fin, B))         # dZ^T[out,B] @ X[B,in]    (TN)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for name, a_mc, b_nc, M, N, K in SHAPES:
        flop = 2.0 * M * N * K
        if name.startswith("fwd"):
            X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
            out = torch.empty(M, N, device=dev)
            blas = lambda: torch.mm(X, W.t(), out=out)
            ours = lambda: ops.gemm(X, W, out=out)
        elif name.startswith("dX"):
            X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev)
            out = torch.empty(M, N, device=dev)
            blas = lambda: torch.mm(X, W, out=out)
            ours = lambda: ops.gemm(X, W, b_nc=True, out=out)
        else:
            dZ = torch.randn(K, M, device=dev); X = torch.randn(K, N, device=dev)
            out = torch.empty(M, N, device=dev)
            blas = lambda: torch.mm(dZ.t(), X, out=out)
            ours = lambda: ops.gemm(dZ, X, a_mc=True, b_nc=True, out=out)
        ref = blas().clone()
        got = ours().clone()
        err = float((ref - got).abs().max() / ref.abs().max())
        to = timeit(ours)
        t = timeit(blas)
        print(json.dumps({"gemm": name, "M": M, "N": N, "K": K, "blas_us": round(t, 2), "blas_tflops": round(flop / t / 1e6, 1),
                          "ours_us": round(to, 2), "ours_tflops": round(flop / to / 1e6, 1), "rel_diff": err}),
              flush=True)


if __name__ == "__main__":
    main()
