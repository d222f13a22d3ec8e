"""CPU-side checks of the C-ABI boundary: the shared library loads and exports exactly the entry points
include/rs_hip.h declares; no compute call is made (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "rs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from recommendsystem_amd import _lib
    names = _declared()
    assert len(names) > 30
    h = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(h, n)]
    assert not missing, missing


def test_python_binding_covers_header():
    from recommendsystem_amd import _lib
    assert sorted(_lib.exported_symbols()) == _declared()
    h = _lib.lib()
    assert h.rs_version() >= 100


def test_host_side_queries():
    from recommendsystem_amd import _lib
    h = _lib.lib()
    assert h.rs_sort_plan_ints(39, 4096) == 39 * (10 * 4096 + 6)
    assert h.rs_sort_max_entries() == 16384
    assert h.rs_sparse_window(16) == 16 and h.rs_sparse_window(1) == 64 and h.rs_sparse_window(20) == 8
    assert h.rs_gemm_workspace_bytes(100, 50, 4) == 4 * 100 * 50 * 4
    assert h.rs_gemm_workspace_bytes(100, 50, 1) == 0
    sk = ctypes.c_int(0)
    assert h.rs_gemm_plan(4096, 1024, 624, 256, ctypes.byref(sk)) == 5 and sk.value == 1
    assert h.rs_gemm_plan(256, 512, 4096, 256, ctypes.byref(sk)) == 3 and sk.value == 8
    assert h.rs_gemm_plan(512, 1024, 4096, 256, ctypes.byref(sk)) == 3 and sk.value == 4


def test_argument_errors_do_not_need_a_gpu():
    from recommendsystem_amd import _lib
    h = _lib.lib()
    rc = h.rs_gather_fwd(None, None, None, None, None, 4, 2, 16, None, None)
    assert rc == -1 and b"null" in h.rs_last_error_string()
    rc = h.rs_gemm_f32(1, 0, 8, 8, 8, ctypes.c_void_p(16), 8, ctypes.c_void_p(16), 8, ctypes.c_void_p(16), 8,
                       None, 0, None, 0, 0, -1, 0, None, 0, None, None)
    assert rc == -1 and b"layout" in h.rs_last_error_string()


def test_cpu_tensors_are_rejected_loudly():
    from recommendsystem_amd import ops
    from recommendsystem_amd._lib import RSError
    with pytest.raises(RSError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))
