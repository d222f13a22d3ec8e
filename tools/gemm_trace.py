"""Per-wave cycle accounting of one GEMM launch (experiment build, RS_GEMM_DBG=64):
start cycle, cycles to the first k-tile, k-loop cycles, total cycles, cycles waiting for DMA / barrier / LDS."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommendsystem_amd import ops, _lib
M, N, K = [int(x) for x in sys.argv[1:4]]
dev = "cuda:0"
a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
out = torch.empty(M, N, device=dev)
tiles = ((M + 63) // 64) * ((N + 63) // 64)
tr = torch.zeros(tiles * 4 * 8, dtype=torch.int32, device=dev)
for it in range(3):
    _lib.call("rs_gemm_f32", 0, 0, M, N, K, _lib.ptr(a), K, _lib.ptr(b), K, _lib.ptr(out), N, None, 0, None, N, 0, 3, 1,
              _lib.ptr(tr), tr.numel() * 4, None, _lib.stream())
torch.cuda.synchronize()
t = tr.view(tiles, 4, 8).cpu().long()
start = t[:, :, 0]
s0 = int(start.min())
rel = (start - s0) & 0xFFFFF
print(f"tiles {tiles}; waves start at (rel. cycles): min {int(rel.min())} median {int(rel.flatten().median())} max {int(rel.max())}")
for name, i in (("to first k-tile", 1), ("k loop", 2), ("total", 3), ("wait dma", 4), ("wait barrier", 5), ("wait lds", 6)):
    v = t[:, :, i].flatten().float()
    print(f"{name:16s} mean {v.mean():9.0f}  min {v.min():9.0f}  max {v.max():9.0f}")
nk = int(t[0, 0, 7])
print(f"k-tiles {nk}; MFMA issue time per wave at 32 cycles: {nk * 32 * 32} cycles")
late = rel.flatten() > 2000
print(f"waves started late (second round): {int(late.sum())}; their k loop mean {t[:, :, 2].flatten()[late].float().mean() if late.any() else 0:.0f}, "
      f"first-round k loop mean {t[:, :, 2].flatten()[~late].float().mean():.0f}")
