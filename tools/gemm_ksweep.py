import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommendsystem_amd import ops
from bench_kernels import timeit
dev="cuda:0"
for (M,N) in [(4096,1024),(4096,512),(4096,256)]:
  for tile in (0,3):
    line=f"M={M} N={N} tile={tile}: "
    for K in (64,128,256,512,1024,2048,4096):
        a=torch.randn(M,K,device=dev); b=torch.randn(N,K,device=dev)
        t=timeit(lambda: ops.gemm(a,b,tile=tile,splitk=1), iters=30)
        line+=f"K={K}:{t:6.1f}us({2*M*N*K/t/1e6:5.1f}TF) "
    print(line, flush=True)
# empty-ish kernel cost: M=64,N=64
a=torch.randn(64,64,device=dev); b=torch.randn(64,64,device=dev)
print("tiny gemm", timeit(lambda: ops.gemm(a,b,tile=3,splitk=1), iters=50))
