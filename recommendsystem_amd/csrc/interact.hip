// K2: feature-interaction kernels on already-gathered embeddings emb[B, F, d] (fp32).
#include "common.h"

namespace {

// ---- FM second order / Bi-Interaction from emb ------------------------------------------------
// One wave per sample; lane k (stride 64) owns embedding component k and walks the F fields.
//   S_k = sum_f e_fk, Q_k = sum_f e_fk^2, bi_k = 0.5*(S_k^2 - Q_k), fm = sum_k bi_k
// (model/FM/FM.py:27-30; the (B, d) vector is the textbook NFM Bi-Interaction pooling).
__global__ __launch_bounds__(256) void fm_fwd_kernel(const float* __restrict__ emb, int B, int F, int d,
                                                     float* __restrict__ sum_out, float* __restrict__ bi_out,
                                                     float* __restrict__ fm_out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* e = emb + (long long)b * F * d;
    float t = 0.f;
    for (int k = lane; k < d; k += 64) {
        float s = 0.f, q = 0.f;
        for (int f = 0; f < F; ++f) { const float v = e[f * d + k]; s += v; q = fmaf(v, v, q); }
        const float bi = 0.5f * (s * s - q);
        if (sum_out) sum_out[(long long)b * d + k] = s;
        if (bi_out) bi_out[(long long)b * d + k] = bi;
        t += bi;
    }
    t = rs_wave_sum(t);
    if (lane == 0 && fm_out) fm_out[b] = t;
}

// demb[b,f,k] = (dfm[b] + dbi[b,k]) * (S[b,k] - emb[b,f,k])
__global__ __launch_bounds__(256) void fm_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ S,
                                                     const float* __restrict__ dfm, const float* __restrict__ dbi,
                                                     int B, int F, int d, float* __restrict__ demb) {
    const long long total = (long long)B * F * d;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int fd = F * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int b = (int)(i / fd), k = (int)(i % d);
        float g = dfm ? dfm[b] : 0.f;
        if (dbi) g += dbi[(long long)b * d + k];
        demb[i] = g * (S[(long long)b * d + k] - emb[i]);
    }
}

int flat_blocks(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int rs_fm_fwd(const float* emb, int B, int F, int d, float* sum_out, float* bi_out, float* fm_out,
              hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 0 && d > 0, "rs_fm_fwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(emb, "rs_fm_fwd: null emb");
    hipLaunchKernelGGL(fm_fwd_kernel, dim3(rs_cdiv(B, 4)), dim3(256), 0, stream, emb, B, F, d, sum_out, bi_out, fm_out);
    RS_CHECK_LAUNCH("rs_fm_fwd");
    return RS_OK;
}

int rs_fm_bwd(const float* emb, const float* S, const float* dfm, const float* dbi, int B, int F, int d, float* demb,
              hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 0 && d > 0, "rs_fm_bwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(emb && S && demb && (dfm || dbi), "rs_fm_bwd: null argument");
    hipLaunchKernelGGL(fm_bwd_kernel, dim3(flat_blocks((long long)B * F * d)), dim3(256), 0, stream, emb, S, dfm, dbi, B,
                       F, d, demb);
    RS_CHECK_LAUNCH("rs_fm_bwd");
    return RS_OK;
}

}  // extern "C"
