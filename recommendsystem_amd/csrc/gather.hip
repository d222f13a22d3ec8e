// K1: fused multi-field embedding gather for gfx950.
//
// One launch serves every field of every sample.  All per-field tables live in one stacked
// [R, d] fp32 array; row_offset[f] is the first row of field f.  The index batch is columnar,
// idx[f*B + b] (int64), i.e. the reference's dict of per-column (B,) tensors laid end to end
// (model/utils/Layers.py:143-151 loops the same columns one nn.Embedding call at a time).
//
// Lane layout inside a 64-lane wave: a sample is served by G*C lanes, G "field groups" times
// C row chunks (C = d/4 float4 chunks when d % 4 == 0, else C = d scalars).  Lane (g, c) walks
// fields f = g, g+G, g+2G, ... so that, for one loop step, the G*C lanes of a sample write
// G consecutive rows of emb_out: G*d*4 contiguous bytes (512 B at d=16, G=8).  Sums over
// fields (FM / Bi-Interaction / LR) stay in registers and are combined across the G groups
// with cross-lane shuffles; nothing is re-read from memory.
#include "gather_body.h"

namespace {

template <int VEC>
__global__ __launch_bounds__(256) void gather_fused_kernel(GatherArgs a) {
    gather_fused_body<VEC>(a, (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6));
}

}  // namespace

extern "C" {

// Fused gather.  Any of the outputs may be null; `table` may be null when only the 1-dim
// (LR) path is wanted, `table1` may be null when only embeddings are wanted.
int rs_gather_lr_fm_fwd_scaled(const int64_t* idx, const int64_t* row_offset, const int64_t* vocab,
                               const float* table, const float* table1, const float* bias,
                               float* emb_out, float* sum_out, float* bi_out, float* lr_out, float* fm_out,
                               int B, int F, int d, const float* scale, int* err_flag, hipStream_t stream);

int rs_gather_lr_fm_fwd(const int64_t* idx, const int64_t* row_offset, const int64_t* vocab,
                        const float* table, const float* table1, const float* bias,
                        float* emb_out, float* sum_out, float* bi_out, float* lr_out, float* fm_out,
                        int B, int F, int d, int* err_flag, hipStream_t stream) {
    return rs_gather_lr_fm_fwd_scaled(idx, row_offset, vocab, table, table1, bias, emb_out, sum_out, bi_out, lr_out,
                                      fm_out, B, F, d, nullptr, err_flag, stream);
}

// rs_gather_lr_fm_fwd on tables kept as w / *scale (scale: a device scalar, null = 1): every looked-up value -- rows,
// 1-dim weights, and through them the sums and the FM term -- is that of the table times *scale.  The bias is not
// scaled.
int rs_gather_lr_fm_fwd_scaled(const int64_t* idx, const int64_t* row_offset, const int64_t* vocab,
                               const float* table, const float* table1, const float* bias,
                               float* emb_out, float* sum_out, float* bi_out, float* lr_out, float* fm_out,
                               int B, int F, int d, const float* scale, int* err_flag, hipStream_t stream) {
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(idx && row_offset && vocab, "rs_gather_lr_fm_fwd: null index arguments");
    RS_CHECK_ARG(B >= 0 && F > 0 && d > 0, "rs_gather_lr_fm_fwd: bad sizes B=%d F=%d d=%d", B, F, d);
    RS_CHECK_ARG(table || table1, "rs_gather_lr_fm_fwd: both tables are null");
    RS_CHECK_ARG(table || !(emb_out || sum_out || bi_out || fm_out),
                 "rs_gather_lr_fm_fwd: embedding outputs requested without a table");
    RS_CHECK_ARG(table1 || !lr_out, "rs_gather_lr_fm_fwd: lr_out requested without table1");
    if (B == 0) return RS_OK;
    GatherArgs a{idx, row_offset, vocab, table, table1, bias, emb_out, sum_out, bi_out, lr_out, fm_out,
                 err_flag, B, F, d, 1, 1};
    a.scale = scale;
    int VEC = 1;
    if (!table) { VEC = 1; a.C = 1; a.G = 8; a.d = 1; }
    else RS_CHECK_ARG(pick_layout(d, &VEC, &a.G, &a.C) == 0,
                      "rs_gather_lr_fm_fwd: unsupported embedding_dim %d (need d%%4==0 && d<=256, or d<=64)", d);
    const int spw = 64 / (a.G * a.C);
    const int waves = rs_cdiv(B, spw);
    const int wpb = 4;
    dim3 grid(rs_cdiv(waves, wpb)), block(64 * wpb);
    if (VEC == 4) hipLaunchKernelGGL(gather_fused_kernel<4>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(gather_fused_kernel<1>, grid, block, 0, stream, a);
    RS_CHECK_LAUNCH("rs_gather_lr_fm_fwd");
    return RS_OK;
}

// Plain multi-field lookup: out[b, f, :] = table[row_offset[f] + idx[f, b], :].
int rs_gather_fwd(const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, const float* table,
                  float* out, int B, int F, int d, int* err_flag, hipStream_t stream) {
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(table && out, "rs_gather_fwd: null table/out");
    return rs_gather_lr_fm_fwd(idx, row_offset, vocab, table, nullptr, nullptr, out, nullptr, nullptr, nullptr,
                               nullptr, B, F, d, err_flag, stream);
}

}  // extern "C"
