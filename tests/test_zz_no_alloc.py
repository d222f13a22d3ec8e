"""rs_colsum_f32_noalloc -- the entry point that allocates nothing (its stage-1 partial sums live in a device array the
library owns) -- against rs_colsum_f32 with a caller-provided workspace: same kernels, so bit-identical results.  Two
calls in a row on the same GPU, the second smaller than the first (the shared array still holds the first call's
partial sums), through the C-ABI."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def colsum_pair(M, N, ld=None, scaled=False, seed=0):
    from recommendsystem_amd import _lib
    from recommendsystem_amd._lib import call, ptr, stream
    h = _lib.lib()
    ld = ld or N
    g = torch.Generator(device="cuda").manual_seed(seed)
    X = torch.randn(M, ld, device="cuda", generator=g)
    scale = torch.randn(M, device="cuda", generator=g) if scaled else None
    ws = torch.empty(max(h.rs_colsum_workspace_bytes(M, N) // 4, 1), device="cuda")
    want = torch.full((N,), float("nan"), device="cuda")
    got = torch.full((N,), float("nan"), device="cuda")
    call("rs_colsum_f32", ptr(X), M, N, ld, ptr(scale), ptr(want), ptr(ws), ws.numel() * 4, stream())
    assert h.rs_colsum_noalloc_prepare() == 0          # (the address look-up of a first use is not what is measured)
    # free device memory as the driver reports it (hipMemGetInfo: sees a hipMalloc / hipFree inside the library, which
    # torch's own counters would not), with torch's allocator quiet: nothing is allocated between the two readings
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    call("rs_colsum_f32_noalloc", ptr(X), M, N, ld, ptr(scale), ptr(got), stream())
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    return got, want, X, scale


def test_noalloc_colsum_matches_workspace_form_on_two_calls_in_a_row():
    got, want, X, _ = colsum_pair(4096, 1024, seed=1)           # 64 x 1024 partial sums
    assert torch.equal(got, want)
    torch.testing.assert_close(got.cpu(), X.double().sum(0).float().cpu(), rtol=1e-5, atol=1e-3)
    got2, want2, X2, s2 = colsum_pair(130, 37, ld=40, scaled=True, seed=2)     # smaller, ragged, unaligned, scaled
    assert torch.equal(got2, want2)
    ref2 = (X2[:, :37].double() * s2.double().view(-1, 1)).sum(0).float()
    torch.testing.assert_close(got2.cpu(), ref2.cpu(), rtol=1e-5, atol=1e-4)
    got3, want3, _, _ = colsum_pair(7, 4, seed=3)
    assert torch.equal(got3, want3)


def test_noalloc_colsum_rejects_what_does_not_fit_and_handles_empty():
    from recommendsystem_amd import _lib
    from recommendsystem_amd._lib import ptr, stream
    h = _lib.lib()
    cap = h.rs_colsum_noalloc_max_floats()
    assert cap == 1 << 20
    M, N = 64 * 1025, 1024                                        # 1025 x 1024 partial sums > 2^20
    X = torch.zeros(1, device="cuda")                             # never read: the call is refused on its sizes
    out = torch.zeros(N, device="cuda")
    rc = h.rs_colsum_f32_noalloc(ptr(X), M, N, N, None, ptr(out), stream())
    assert rc == -1 and b"exceed" in h.rs_last_error_string()
    out = torch.ones(8, device="cuda")
    rc = h.rs_colsum_f32_noalloc(None, 0, 8, 8, None, ptr(out), stream())     # no rows: zeros
    torch.cuda.synchronize()
    assert rc == 0 and torch.count_nonzero(out).item() == 0


def test_noalloc_colsum_under_graph_capture():
    """After rs_colsum_noalloc_prepare() (or one ordinary call) on the device, the call is a pair of kernel launches
    and nothing else: captured in a HIP graph -- where an allocation or a synchronising runtime call would fail the
    capture -- and replayed twice on changed inputs, it gives rs_colsum_f32's sums bit for bit."""
    from recommendsystem_amd import _lib
    from recommendsystem_amd._lib import call, ptr, stream
    h = _lib.lib()
    M, N = 1000, 96
    X = torch.randn(M, N, device="cuda")
    got = torch.zeros(N, device="cuda")
    want = torch.zeros(N, device="cuda")
    ws = torch.empty(max(h.rs_colsum_workspace_bytes(M, N) // 4, 1), device="cuda")
    assert h.rs_colsum_noalloc_prepare() == 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            call("rs_colsum_f32_noalloc", ptr(X), M, N, N, None, ptr(got), stream())
    for seed in (1, 2):
        X.copy_(torch.randn(M, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed)))
        g.replay()
        call("rs_colsum_f32", ptr(X), M, N, N, None, ptr(want), ptr(ws), ws.numel() * 4, stream())
        torch.cuda.synchronize()
        assert torch.equal(got, want)
