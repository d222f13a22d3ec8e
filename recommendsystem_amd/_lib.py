"""ctypes binding of librs_hip.so, the C-ABI boundary of the hot path.

The library is the product: if it is missing or a symbol cannot be resolved this module raises.
There is no CPU or eager-PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RS_HIP_LIB") or os.path.join(_HERE, "lib", "librs_hip.so")   # RS_HIP_LIB: experiment builds

c_i64p = C.c_void_p
c_f32p = C.c_void_p
c_i32p = C.c_void_p
c_stream = C.c_void_p
I = C.c_int
F = C.c_float
L = C.c_longlong



class GemmDesc(C.Structure):
    """rs_gemm_desc of include/rs_hip.h: one product of rs_gemm_f32_group."""
    _fields_ = [("a_mc", I), ("b_nc", I), ("M", I), ("N", I), ("K", I),
                ("A", C.c_void_p), ("lda", I),
                ("B", C.c_void_p), ("ldb", I),
                ("C", C.c_void_p), ("ldc", I),
                ("bias", C.c_void_p), ("act", I),
                ("mulsrc", C.c_void_p), ("ldm", I), ("mul_act", I),
                ("splitk", I),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("colsum_part", C.c_void_p),
                ("drop_p", C.c_float), ("drop_salt", C.c_uint), ("drop_seed", C.c_ulonglong),
                ("drop_call_host", C.c_ulonglong), ("drop_call_dev", C.c_void_p)]


# name -> argtypes (every entry point returns int, except where noted below)
SZ = C.c_size_t
_SIGS = {
    # gather.hip
    "rs_gather_fwd": [c_i64p, c_i64p, c_i64p, c_f32p, c_f32p, I, I, I, c_i32p, c_stream],
    "rs_gather_lr_fm_fwd": [c_i64p, c_i64p, c_i64p, c_f32p, c_f32p, c_f32p,
                            c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I, I, I, c_i32p, c_stream],
    "rs_gather_lr_fm_fwd_scaled": [c_i64p, c_i64p, c_i64p, c_f32p, c_f32p, c_f32p,
                                   c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_i32p, c_stream],
    # gemm.hip
    "rs_gemm_f32": [I, I, I, I, I, c_f32p, I, c_f32p, I, c_f32p, I, c_f32p, I, c_f32p, I, I,
                    I, I, c_f32p, SZ, c_f32p, c_stream],
    "rs_gemm_f32_drop": [I, I, I, I, I, c_f32p, I, c_f32p, I, c_f32p, I, c_f32p, I, c_f32p, I, I,
                         I, I, c_f32p, SZ, c_f32p, F, C.c_ulonglong, C.c_uint, C.c_ulonglong, c_i64p, c_stream],
    "rs_gemm_f32_acc": [I, I, I, I, I, c_f32p, I, c_f32p, I, c_f32p, I, c_f32p, I, I, I, c_f32p, SZ, c_stream],
    "rs_colsum_f32": [c_f32p, I, I, I, c_f32p, c_f32p, c_f32p, SZ, c_stream],
    "rs_colsum_partials": [c_f32p, I, I, I, c_f32p, c_f32p, c_stream],
    "rs_colsum_f32_noalloc": [c_f32p, I, I, I, c_f32p, c_f32p, c_stream],
    "rs_gemm_f32_slabs": [I, I, I, I, I, c_f32p, I, c_f32p, I, I, I, c_f32p, SZ, c_stream],
    "rs_gemm_f32_group": [C.c_void_p, I, I, c_stream],
    "rs_gemm_f32_group_sort": [C.c_void_p, I, c_i64p, c_i32p, c_i32p, c_i32p, I, I, c_i64p, c_i64p, c_i32p, I, I, I, I,
                               c_stream],
    "rs_splitk_reduce_multi": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, I, c_stream],
    "rs_colsum_finish_multi": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, I, c_stream],
    # sparse.hip
    "rs_sparse_bwd_sort": [c_i64p, c_i32p, c_i32p, c_i32p, I, I, I, I, I, I, c_i64p, c_i64p, c_i32p, c_stream],
    "rs_sparse_bwd_merge_lists": [c_i64p, c_i32p, c_i32p, c_i32p, I, I, I, I, I, I, c_i64p, c_i64p, c_i32p, c_stream],
    "rs_sort_gather": [c_i64p, c_i32p, c_i32p, c_i32p, I, I, c_i64p, c_i64p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p,
                       c_f32p, c_f32p, c_f32p, c_f32p, I, I, I, c_i32p, I, c_stream],
    "rs_sparse_bwd_reduce": [c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I,
                             c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I, I, c_stream],
    "rs_sparse_bwd_reduce_sumsq": [c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I,
                                   c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, L, c_f32p, I,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, I,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, I, I, I, c_stream],
    "rs_sparse_bwd_reduce_routed": [c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I,
                                    c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, L, c_f32p, I,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, I,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, I, I, I, c_i32p, I, I, c_stream],
    "rs_sparse_dense_update": [c_i32p, I, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                               c_f32p, c_f32p, c_f32p, c_f32p, L, c_f32p, c_f32p, c_i32p, c_i32p, c_i64p, c_stream],
    "rs_sparse_update": [c_i32p, I, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                         c_f32p, c_f32p, c_i32p, c_i32p, c_i64p, c_stream],
    "rs_sparse_scatter_dense": [c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_stream],
    "rs_dense_update": [c_f32p, c_f32p, c_f32p, c_f32p, L, I, c_f32p, c_f32p, c_stream],
    "rs_sumsq_partials": [c_f32p, L, c_f32p, I, c_stream],
    "rs_clip_coef": [c_f32p, I, F, c_f32p, c_stream],
    "rs_opt_prepare": [c_i64p, c_f32p, c_stream],
    "rs_clip_and_prepare": [c_f32p, I, F, c_f32p, c_i64p, c_f32p, c_stream],
    "rs_sparse_touched_dot": [c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I, c_stream],
    "rs_clip_and_prepare_l2": [c_f32p, I, I, c_f32p, I, F, F, C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64p,
                               c_f32p, c_stream],
    "rs_scale_inplace": [c_f32p, L, C.c_void_p, c_stream],
    # rowstate.hip
    "rs_adam_bc_table": [c_f32p, I, c_f32p, c_stream],
    "rs_adam_catchup": [c_i64p, c_i64p, c_i64p, I, I, I, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, c_i32p,
                        I, c_i64p, I, c_f32p, c_f32p, I, I, c_stream],
    "rs_adam_catchup_lists": [c_i64p, c_i32p, c_i32p, I, I, c_i64p, c_i64p, I, I, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p,
                              c_f32p, c_i32p, I, c_i64p, I, c_f32p, c_f32p, I, I, c_stream],
    "rs_sort_catchup": [c_i64p, c_i32p, c_i32p, c_i32p, I, I, c_i64p, c_i64p, c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_i32p,
                        c_f32p, c_f32p, c_f32p, c_i32p, I, c_i64p, c_f32p, c_f32p, I, I, I, c_stream],
    "rs_adam_catchup_plan": [c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, c_i32p, I, c_i64p,
                             c_f32p, c_f32p, I, I, I, c_stream],
    "rs_adam_catchup_serve": [c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, c_i32p, I, c_i64p,
                              c_f32p, c_f32p, I, I, c_f32p, I, I, I, c_stream],
    "rs_sparse_dense_update_ahead": [c_i32p, I, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                     c_f32p, c_f32p, c_f32p, c_f32p, L, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, c_i64p,
                                     c_f32p, I, I, c_stream],
    "rs_adam_flush": [c_f32p, c_f32p, c_f32p, c_i32p, L, I, I, c_i64p, c_f32p, c_f32p, I, I, c_stream],
    # elementwise.hip
    "rs_bce_head": [c_f32p, c_f32p, c_f32p, c_f32p, c_i64p, I, c_f32p, c_f32p, c_f32p, c_f32p, F,
                    c_f32p, c_f32p, I, c_f32p, c_f32p, c_stream],
    "rs_rowdot_fwd": [c_f32p, I, I, c_f32p, c_f32p, I, c_f32p, c_stream],
    "rs_rowdot_bwd_dx": [c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_stream],
    "rs_head_fwd_bce": [c_f32p, I, I, c_f32p, c_f32p, I, c_f32p, c_f32p, c_f32p, c_i64p, F, c_f32p, c_f32p, c_f32p,
                        c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_stream],
    "rs_head_fwd_bwd": [c_f32p, I, I, c_f32p, c_f32p, I, c_f32p, c_f32p, c_f32p, c_i64p, F, c_f32p, c_f32p, c_f32p,
                        c_f32p, I, c_f32p, c_f32p, c_stream],
    "rs_head_bwd": [c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_f32p, c_stream],
    "rs_rowdot2_fwd": [c_f32p, I, c_f32p, I, I, c_f32p, c_f32p, I, c_f32p, c_stream],
    "rs_head2_bwd": [c_f32p, c_f32p, c_f32p, I, c_f32p, I, I, I, c_f32p, c_f32p, c_f32p, c_stream],
    "rs_act_fwd": [c_f32p, c_f32p, L, I, c_f32p, c_stream],
    "rs_act_bwd": [c_f32p, c_f32p, c_f32p, c_f32p, L, I, c_f32p, c_f32p, c_stream],
    "rs_bn_fwd": [c_f32p, I, I, c_f32p, c_f32p, F, I, I, F, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream],
    "rs_bn_bwd": [c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_f32p, F, c_f32p, c_f32p, c_f32p, c_stream],
    "rs_bn_stats": [c_f32p, I, I, c_f32p, c_f32p, c_stream],
    "rs_bn_bwd_reduce": [c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_f32p, c_stream],
    "rs_bn_bwd_apply": [c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_f32p, F, c_f32p, c_f32p, I, c_f32p, c_stream],
    "rs_ln_fwd": [c_f32p, I, I, c_f32p, c_f32p, F, I, c_f32p, c_f32p, c_f32p, c_stream],
    "rs_ln_bwd": [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_f32p, c_f32p, c_stream],
    "rs_add_f32": [c_f32p, c_f32p, c_f32p, L, c_stream],
    "rs_dropout_apply": [c_f32p, c_f32p, L, F, C.c_ulonglong, C.c_uint, C.c_ulonglong, c_i64p, c_stream],
    "rs_axpy_f32": [F, c_f32p, c_f32p, L, c_stream],
    "rs_loss_add_reg": [c_f32p, c_f32p, c_i32p, c_f32p, I, c_stream],
    # shard.hip
    "rs_shard_route": [c_i32p, I, I, c_i64p, I, c_i32p, c_i32p, I, c_i64p, c_i32p, c_i32p, c_stream],
    "rs_shard_gather": [c_i64p, c_i64p, c_i64p, I, I, c_i32p, c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, I, I,
                        c_i32p, c_stream],
    "rs_shard_lookup": [c_i32p, c_f32p, I, c_f32p, I, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I, I, I, c_stream],
    "rs_shard_expand": [c_i32p, c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_stream],
    "rs_shard_pack": [c_i32p, c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, I, I, c_stream],
    "rs_shard_counts": [c_i64p, I, I, c_i32p, c_i32p, I, c_i32p, c_i32p, c_i32p, I, c_i32p, I, I, c_stream],
    # interact.hip
    "rs_fm_fwd": [c_f32p, I, I, I, c_f32p, c_f32p, c_f32p, c_stream],
    "rs_fm_bwd": [c_f32p, c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_stream],
    "rs_pnn_inner_fwd": [c_f32p, I, I, I, c_f32p, c_stream],
    "rs_pnn_inner_bwd": [c_f32p, c_f32p, I, I, I, c_f32p, c_stream],
    "rs_pnn_inner_cat_fwd": [c_f32p, I, I, I, c_f32p, L, c_stream],
    "rs_pnn_inner_cat_bwd": [c_f32p, c_f32p, L, I, I, I, c_f32p, c_stream],
    "rs_pnn_outer_fwd": [c_f32p, I, I, I, c_f32p, c_stream],
    "rs_pnn_outer_bwd": [c_f32p, c_f32p, I, I, I, c_f32p, c_stream],
    "rs_pnn_outer_fwd_range": [c_f32p, I, I, I, I, I, c_f32p, L, c_stream],
    "rs_pnn_outer_bwd_range": [c_f32p, c_f32p, L, I, I, I, I, I, I, c_f32p, c_stream],
    "rs_pnn_outer_mlp_fwd": [c_f32p, c_f32p, L, I, I, I, I, c_f32p, c_f32p, I, c_f32p, c_f32p, SZ, L, c_stream],
    "rs_pnn_outer_mlp_bwd": [c_f32p, c_f32p, L, c_f32p, I, I, I, I, c_f32p, I, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, SZ,
                             L, c_stream],
    "rs_dcn_cross_fwd": [c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_f32p, c_stream],
    "rs_dcn_cross_bwd": [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_f32p, I, c_stream],
    "rs_dcn_cross_bwd_add": [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_f32p, c_f32p, I, c_stream],
    "rs_nfm_bi_fwd": [c_f32p, c_f32p, I, I, I, c_f32p, c_f32p, c_stream],
    "rs_nfm_bi_bwd": [c_f32p, c_f32p, c_f32p, c_f32p, I, I, I, c_f32p, c_f32p, I, c_stream],
    "rs_afm_fwd": [c_f32p, I, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream],
    "rs_afm_bwd": [c_f32p, c_f32p, I, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, I, c_stream],
    "rs_ffm_fwd": [c_i64p, c_i64p, c_i64p, c_f32p, I, I, I, c_f32p, c_i32p, c_stream],
    "rs_ffm_bwd": [c_i32p, c_i64p, c_i64p, c_i64p, c_f32p, c_f32p, I, I, I, I, c_f32p, c_stream],
    "rs_ffm_rows_fwd": [c_f32p, I, I, I, c_f32p, c_stream],
    "rs_ffm_rows_bwd": [c_f32p, c_f32p, I, I, I, c_f32p, c_stream],
    "rs_sparse_bwd_reduce_wide": [c_i32p, I, I, I, c_f32p, c_f32p, c_f32p, c_f32p, c_stream],
    "rs_ffm_bwd_rows": [c_i32p, c_i64p, c_i64p, c_i64p, c_f32p, c_f32p, I, I, I, I, c_f32p, c_f32p, c_f32p, c_stream],
}
# entry points that do not follow the "returns int status" convention: name -> (restype, argtypes)
_PLAIN = {
    "rs_version": (I, []),
    "rs_last_error_string": (C.c_char_p, []),
    "rs_device_cu_count": (I, []),
    "rs_gemm_workspace_bytes": (SZ, [I, I, I]),
    "rs_gemm_plan": (I, [I, I, I, I, C.POINTER(I)]),
    "rs_gemm_colsum_rows": (I, [I, I]),
    "rs_head_bwd_rows": (I, [I]),
    "rs_head_workspace_floats": (SZ, [I, I]),
    "rs_colsum_workspace_bytes": (SZ, [I, I]),
    "rs_colsum_partial_rows": (I, [I]),
    "rs_colsum_noalloc_max_floats": (SZ, []),
    "rs_colsum_noalloc_prepare": (I, []),
    "rs_sort_plan_ints": (SZ, [I, I]),
    "rs_pnn_outer_mlp_workspace_floats": (SZ, [I, I, I, I, I, I, L]),
    "rs_pnn_outer_mlp_blocks": (I, [I, I, I, L]),
    "rs_pnn_inner_cat_ok": (I, [I, I]),
    "rs_sort_max_entries": (I, []),
    "rs_group_sort_max_entries": (I, []),
    "rs_merge_max_entries": (I, []),
    "rs_sort_split_max_entries": (I, []),
    "rs_sparse_bwd_num_partials": (I, [I, I, I]),
    "rs_sparse_window": (I, [I]),
    "rs_sparse_touched_dot_groups": (I, [I, I]),
    "rs_act_bwd_num_partials": (I, [L]),
    "rs_interact_partial_rows": (I, [I]),
    "rs_ffm_bwd_num_partials": (I, [I, I]),
    "rs_sparse_bwd_wide_num_partials": (I, [I, I]),
}

_lib = None


class RSError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RSError(f"{LIB_PATH} not found: build it with `python -m recommendsystem_amd.build` "
                      "(there is no fallback path)")
    h = C.CDLL(LIB_PATH)
    for name, (res, args) in _PLAIN.items():
        fn = getattr(h, name)
        fn.restype = res
        fn.argtypes = args
    for name, args in _SIGS.items():
        fn = getattr(h, name)          # AttributeError if the symbol is missing: loud by design
        fn.restype = I
        fn.argtypes = args
    _lib = h
    return h


def exported_symbols():
    return [*_PLAIN.keys(), *_SIGS.keys()]


def check(rc: int, what: str):
    if rc != 0:
        raise RSError(f"{what} failed (code {rc}): {lib().rs_last_error_string().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream():
    # the raw handle of torch's current stream without building a torch.cuda.Stream object (eager steps ask ~30 times)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RSError("recommendsystem_amd runs on MI355X only: got a CPU tensor "
                          "(the package has no CPU fallback)")


n_calls = 0      # entry-point calls so far (lets a graph recorder tell an empty segment from one with launches)


_fn_cache = {}


def call(name: str, *args):
    global n_calls
    n_calls += 1
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        check(rc, name)
