// Small fused elementwise / reduction kernels around the GEMMs: loss head, normalisation layers.
#include "common.h"
#include "philox.h"

namespace {

// logit = z0 + z1 + z2 (null parts skipped); p = sigmoid(logit);
// loss = mean(-(y*max(log p,-100) + (1-y)*max(log1p(-p),-100)))    [F.binary_cross_entropy,
// model/utils/BaseModel.py:75,145 applied to sigmoid outputs as model/LR/LR.py:32 does]
// dz = dloss/dlogit following autograd's two steps (BCE backward then sigmoid backward).
// Optional extras for the fused step (all sums in fixed order, one workgroup):
//   sum_dz_out[0]   = sum_b dz[b]                     (gradient of a bias added to the logit)
//   dzh_out[b]      = dz[b] * act'(hlast[b])          (gradient at the pre-activation of a 1-wide last
//   sum_dzh_out[0]  = sum_b dzh_out[b]                 layer whose output hlast is one of the parts)
__global__ __launch_bounds__(1024) void bce_head_kernel(const float* __restrict__ z0, const float* __restrict__ z1,
                                                        const float* __restrict__ z2, const float* __restrict__ yf,
                                                        const int64_t* __restrict__ yi, int B,
                                                        float* __restrict__ logit_out, float* __restrict__ prob_out,
                                                        float* __restrict__ loss_out, float* __restrict__ dz_out,
                                                        float grad_scale, float* __restrict__ sum_dz_out,
                                                        const float* __restrict__ hlast, int act_last,
                                                        float* __restrict__ dzh_out, float* __restrict__ sum_dzh_out) {
    __shared__ float red[16];
    float s = 0.f, sdz = 0.f, sdzh = 0.f;
    const float invB = 1.f / (float)B;
    for (int b = threadIdx.x; b < B; b += 1024) {
        float z = z0[b];
        if (z1) z += z1[b];
        if (z2) z += z2[b];
        const float y = yf ? yf[b] : (float)yi[b];
        const float p = 1.f / (1.f + expf(-z));
        const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(log1pf(-p), -100.f);
        s += -(y * lp + (1.f - y) * l1p);
        if (logit_out) logit_out[b] = z;
        if (prob_out) prob_out[b] = p;
        const float gp = (p - y) / fmaxf((1.f - p) * p, 1e-12f) * invB * grad_scale;
        const float dz = gp * ((1.f - p) * p);
        if (dz_out) dz_out[b] = dz;
        sdz += dz;
        if (hlast) {
            const float h = hlast[b];
            const float dzh = dz * rs_act_grad(act_last, h, h, 0.f);
            if (dzh_out) dzh_out[b] = dzh;
            sdzh += dzh;
        }
    }
    const float tot = rs_block_sum<1024>(s, red);
    if (threadIdx.x == 0 && loss_out) loss_out[0] = tot * invB;
    if (sum_dz_out) {
        const float t = rs_block_sum<1024>(sdz, red);
        if (threadIdx.x == 0) sum_dz_out[0] = t;
    }
    if (sum_dzh_out) {
        const float t = rs_block_sum<1024>(sdzh, red);
        if (threadIdx.x == 0) sum_dzh_out[0] = t;
    }
}

// ---- 1-wide linear layer (the MLP head, model/DeepFM/DeepFM.py:35): h[m] = act(x[m,:].w + b) -----
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ X, int M, int K,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         int act, float* __restrict__ h) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* x = X + (long long)m * K;
    float s = 0.f;
    if ((K & 3) == 0) {
        for (int k = lane * 4; k < K; k += 256) {
            const float4 a = *reinterpret_cast<const float4*>(x + k);
            const float4 b = *reinterpret_cast<const float4*>(w + k);
            s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
        }
    } else {
        for (int k = lane; k < K; k += 64) s = fmaf(x[k], w[k], s);
    }
    s = rs_wave_sum(s);
    if (lane == 0) h[m] = rs_act(act, s + (bias ? bias[0] : 0.f), 0.f);
}

// The same for an input given as two column blocks, h[m] = act(x1[m,:].w[:K1] + x2[m,:].w[K1:] + b): a 1-wide Linear
// over cat([x1, x2], 1) (model/DCN/DCN.py:31,52) without the concatenation.
__global__ __launch_bounds__(256) void rowdot2_fwd_kernel(const float* __restrict__ X1, int K1, const float* __restrict__ X2,
                                                          int K2, int M, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int act, float* __restrict__ h) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* x1 = X1 + (long long)m * K1;
    const float* x2 = X2 + (long long)m * K2;
    float s = 0.f;
    if (((K1 | K2) & 3) == 0) {
        for (int k = lane * 4; k < K1; k += 256) {
            const float4 a = *reinterpret_cast<const float4*>(x1 + k);
            const float4 b = *reinterpret_cast<const float4*>(w + k);
            s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
        }
        for (int k = lane * 4; k < K2; k += 256) {
            const float4 a = *reinterpret_cast<const float4*>(x2 + k);
            const float4 b = *reinterpret_cast<const float4*>(w + K1 + k);
            s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
        }
    } else {
        for (int k = lane; k < K1; k += 64) s = fmaf(x1[k], w[k], s);
        for (int k = lane; k < K2; k += 64) s = fmaf(x2[k], w[K1 + k], s);
    }
    s = rs_wave_sum(s);
    if (lane == 0) h[m] = rs_act(act, s + (bias ? bias[0] : 0.f), 0.f);
}

// dX[m,k] = dz[m] * w[k] * act'_prev(X[m,k])   (X is the previous block's output; mul_act = its activation)
__global__ __launch_bounds__(256) void rowdot_bwd_dx_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                            const float* __restrict__ X, int mul_act, int M, int K,
                                                            float* __restrict__ dX) {
    const long long total = (long long)M * K;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int m = (int)(i / K), k = (int)(i % K);
        float v = dz[m] * w[k];
        if (mul_act) { const float hh = X[i]; v *= rs_act_grad(mul_act, hh, hh, 0.f); }
        dX[i] = v;
    }
}


// ---- fused head: 1-wide last layer + loss head in one launch, and its backward -----------------------
// Forward (model/DeepFM/DeepFM.py:35,54 + BaseModel.py:145): h[m] = act(X[m,:].w + b), logit = z0 + z1 + h,
// BCE loss (mean), dz = dL/dlogit * grad_scale, dzh = dz * act'(h).  A workgroup takes 16 rows: each of its
// four waves forms four row dot products, then 16 lanes do the per-row loss arithmetic.  The per-workgroup
// sums (loss, sum dz, sum dzh) are added in workgroup order by the last workgroup to arrive.
constexpr int HEAD_ROWS = 16;
__global__ __launch_bounds__(256) void head_fwd_bce_kernel(const float* __restrict__ X, int B, int K,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           int act, const float* __restrict__ z0,
                                                           const float* __restrict__ z1, const float* __restrict__ yf,
                                                           const int64_t* __restrict__ yi, float grad_scale,
                                                           float* __restrict__ h_out, float* __restrict__ loss_out,
                                                           float* __restrict__ dz_out, float* __restrict__ sum_dz_out,
                                                           float* __restrict__ dzh_out, float* __restrict__ sum_dzh_out,
                                                           float* __restrict__ part /*[grid,4]*/, int* sem) {
    __shared__ float hrow[HEAD_ROWS];
    __shared__ int s_flag;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = blockIdx.x * HEAD_ROWS;
    const float b0 = bias ? bias[0] : 0.f;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane * 4; k < K; k += 256) {
        const float4 c = *reinterpret_cast<const float4*>(w + k);
        float4 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = r0 + wv * 4 + i;
            a[i] = m < B ? *reinterpret_cast<const float4*>(X + (long long)m * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i] = fmaf(a[i].x, c.x, s[i]); s[i] = fmaf(a[i].y, c.y, s[i]);
            s[i] = fmaf(a[i].z, c.z, s[i]); s[i] = fmaf(a[i].w, c.w, s[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float t = rs_wave_sum(s[i]);
        if (lane == 0) hrow[wv * 4 + i] = rs_act(act, t + b0, 0.f);
    }
    __syncthreads();
    if (wv == 0) {
        const int m = r0 + lane;
        float l = 0.f, dz = 0.f, dzh = 0.f;
        if (lane < HEAD_ROWS && m < B) {
            const float h = hrow[lane];
            // the parts add up in model order (first-order + second-order + deep), as in rs_bce_head
            float z = h;
            if (z0 && z1) z = (z0[m] + z1[m]) + h;
            else if (z0) z = z0[m] + h;
            const float y = yf ? yf[m] : (float)yi[m];
            const float p = 1.f / (1.f + expf(-z));
            const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(log1pf(-p), -100.f);
            l = -(y * lp + (1.f - y) * l1p);
            const float gp = (p - y) / fmaxf((1.f - p) * p, 1e-12f) * (1.f / (float)B) * grad_scale;
            dz = gp * ((1.f - p) * p);
            dzh = dz * rs_act_grad(act, h, h, 0.f);
            if (h_out) h_out[m] = h;
            if (dz_out) dz_out[m] = dz;
            if (dzh_out) dzh_out[m] = dzh;
        }
        l = rs_wave_sum(l); dz = rs_wave_sum(dz); dzh = rs_wave_sum(dzh);
        if (sem == nullptr) {
            // deferred sums: the caller adds the columns of part (loss already scaled by 1/B) in a later launch
            if (lane == 0) {
                part[blockIdx.x * 4 + 0] = l * (1.f / (float)B);
                part[blockIdx.x * 4 + 1] = dz;
                part[blockIdx.x * 4 + 2] = dzh;
            }
            return;
        }
        if (lane == 0) {
            rs_st_agent(part + blockIdx.x * 4 + 0, l);
            rs_st_agent(part + blockIdx.x * 4 + 1, dz);
            rs_st_agent(part + blockIdx.x * 4 + 2, dzh);
        }
    }
    if (sem == nullptr) return;                          // uniform over the launch
    if (!rs_last_arrival(sem, gridDim.x, &s_flag)) return;
    if (wv == 0) {
        float l = 0.f, dz = 0.f, dzh = 0.f;
        for (int g = lane; g < (int)gridDim.x; g += 64) {
            l += rs_ld_agent(part + g * 4 + 0); dz += rs_ld_agent(part + g * 4 + 1); dzh += rs_ld_agent(part + g * 4 + 2);
        }
        l = rs_wave_sum(l); dz = rs_wave_sum(dz); dzh = rs_wave_sum(dzh);
        if (lane == 0) {
            if (loss_out) loss_out[0] = l * (1.f / (float)B);
            if (sum_dz_out) sum_dz_out[0] = dz;
            if (sum_dzh_out) sum_dzh_out[0] = dzh;
        }
    }
}

// Backward of the 1-wide layer:
//   dX[m,k] = dzh[m] * w[k] * act'_prev(X[m,k])    dW[k] = sum_m dzh[m] X[m,k]    dbp[k] = sum_m dX[m,k]
// (dbp is the bias gradient of the previous block when its act' is fused, i.e. dX is that block's dZ).
// A workgroup takes 16 rows, thread t the columns t, t+256, ... (at most 4), and leaves one row of partial
// column sums [dW | dbp]; the caller adds the partial rows in row order (rs_colsum_finish_multi).
constexpr int HEAD_KPT = 4;
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dzh, const float* __restrict__ w,
                                                       const float* __restrict__ X, int mul_act, int B, int K,
                                                       float* __restrict__ dX, float* __restrict__ part /*[grid,2,K]*/) {
    __shared__ float sdz[HEAD_ROWS];
    const int r0 = blockIdx.x * HEAD_ROWS;
    if (threadIdx.x < HEAD_ROWS) sdz[threadIdx.x] = r0 + threadIdx.x < B ? dzh[r0 + threadIdx.x] : 0.f;
    __syncthreads();
    float* prow = part + (long long)blockIdx.x * 2 * K;
#pragma unroll
    for (int u = 0; u < HEAD_KPT; ++u) {
        const int k = threadIdx.x + u * 256;
        if (k >= K) break;
        const float wk = w[k];
        float x[HEAD_ROWS];
#pragma unroll
        for (int r = 0; r < HEAD_ROWS; ++r) x[r] = r0 + r < B ? X[(long long)(r0 + r) * K + k] : 0.f;
        float aw = 0.f, ab = 0.f;
#pragma unroll
        for (int r = 0; r < HEAD_ROWS; ++r) {
            const float g = sdz[r];
            float v = g * wk;
            if (mul_act) v *= rs_act_grad(mul_act, x[r], x[r], 0.f);
            if (r0 + r < B) dX[(long long)(r0 + r) * K + k] = v;
            aw = fmaf(g, x[r], aw);
            ab += v;
        }
        prow[k] = aw;
        prow[K + k] = ab;
    }
}

// head_bwd_kernel for an input given as two column blocks X1 [B,K1], X2 [B,K2] (rowdot2_fwd_kernel): column k < K1 of
// the concatenation lives in X1 / dX1, the others in X2 / dX2; mul_act2 (the activation whose output X2 is) applies
// to the second block only.  part: [grid, 2, K1 + K2].
__global__ __launch_bounds__(256) void head2_bwd_kernel(const float* __restrict__ dzh, const float* __restrict__ w,
                                                        const float* __restrict__ X1, int K1, const float* __restrict__ X2,
                                                        int K2, int mul_act2, int B, float* __restrict__ dX1,
                                                        float* __restrict__ dX2, float* __restrict__ part) {
    __shared__ float sdz[HEAD_ROWS];
    const int r0 = blockIdx.x * HEAD_ROWS, K = K1 + K2;
    if (threadIdx.x < HEAD_ROWS) sdz[threadIdx.x] = r0 + threadIdx.x < B ? dzh[r0 + threadIdx.x] : 0.f;
    __syncthreads();
    float* prow = part + (long long)blockIdx.x * 2 * K;
#pragma unroll
    for (int u = 0; u < HEAD_KPT; ++u) {
        const int k = threadIdx.x + u * 256;
        if (k >= K) break;
        const bool second = k >= K1;
        const float* X = second ? X2 : X1;
        float* dX = second ? dX2 : dX1;
        const int kk = second ? k - K1 : k, Kx = second ? K2 : K1;
        const int ma = second ? mul_act2 : 0;
        const float wk = w[k];
        float x[HEAD_ROWS];
#pragma unroll
        for (int r = 0; r < HEAD_ROWS; ++r) x[r] = r0 + r < B ? X[(long long)(r0 + r) * Kx + kk] : 0.f;
        float aw = 0.f, ab = 0.f;
#pragma unroll
        for (int r = 0; r < HEAD_ROWS; ++r) {
            const float g = sdz[r];
            float v = g * wk;
            if (ma) v *= rs_act_grad(ma, x[r], x[r], 0.f);
            if (r0 + r < B) dX[(long long)(r0 + r) * Kx + kk] = v;
            aw = fmaf(g, x[r], aw);
            ab += v;
        }
        prow[k] = aw;
        prow[K + k] = ab;
    }
}

// head_fwd_bce_kernel and head_bwd_kernel in one launch: the 16 rows a workgroup has just used for the dot products
// are still in its registers when their dzh is known, so dX, and the workgroup's partial rows of dW and dbp, come
// out without reading X again.  Batch sums: left as partials (part4, see head_fwd_bce_kernel with sem == nullptr).
// KI = ceil(K / 256) float4 columns per lane, K <= 1024.
template <int KI>
__global__ __launch_bounds__(256) void head_fwd_bwd_kernel(const float* __restrict__ X, int B, int K,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           int act, const float* __restrict__ z0,
                                                           const float* __restrict__ z1, const float* __restrict__ yf,
                                                           const int64_t* __restrict__ yi, float grad_scale,
                                                           float* __restrict__ h_out, float* __restrict__ dz_out,
                                                           float* __restrict__ dzh_out, float* __restrict__ part4,
                                                           int mul_act, float* __restrict__ dX,
                                                           float* __restrict__ prow_all /*[grid,2,K]*/) {
    extern __shared__ float wred[];                     // [4 waves][2K]
    __shared__ float hrow[HEAD_ROWS], sdzh[HEAD_ROWS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = blockIdx.x * HEAD_ROWS;
    const float b0 = bias ? bias[0] : 0.f;
    float4 a[KI][4], c[KI];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < KI; ++it) {
        const int k = lane * 4 + it * 256;
        const bool in = k < K;
        c[it] = in ? *reinterpret_cast<const float4*>(w + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = r0 + wv * 4 + i;
            a[it][i] = (in && m < B) ? *reinterpret_cast<const float4*>(X + (long long)m * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i] = fmaf(a[it][i].x, c[it].x, s[i]); s[i] = fmaf(a[it][i].y, c[it].y, s[i]);
            s[i] = fmaf(a[it][i].z, c[it].z, s[i]); s[i] = fmaf(a[it][i].w, c[it].w, s[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float t = rs_wave_sum(s[i]);
        if (lane == 0) hrow[wv * 4 + i] = rs_act(act, t + b0, 0.f);
    }
    __syncthreads();
    if (wv == 0) {
        const int m = r0 + lane;
        float l = 0.f, dz = 0.f, dzh = 0.f;
        if (lane < HEAD_ROWS && m < B) {
            const float h = hrow[lane];
            float z = h;
            if (z0 && z1) z = (z0[m] + z1[m]) + h;
            else if (z0) z = z0[m] + h;
            const float y = yf ? yf[m] : (float)yi[m];
            const float p = 1.f / (1.f + expf(-z));
            const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(log1pf(-p), -100.f);
            l = -(y * lp + (1.f - y) * l1p);
            const float gp = (p - y) / fmaxf((1.f - p) * p, 1e-12f) * (1.f / (float)B) * grad_scale;
            dz = gp * ((1.f - p) * p);
            dzh = dz * rs_act_grad(act, h, h, 0.f);
            if (h_out) h_out[m] = h;
            if (dz_out) dz_out[m] = dz;
            if (dzh_out) dzh_out[m] = dzh;
        }
        if (lane < HEAD_ROWS) sdzh[lane] = dzh;
        l = rs_wave_sum(l); dz = rs_wave_sum(dz); dzh = rs_wave_sum(dzh);
        if (lane == 0) {
            part4[blockIdx.x * 4 + 0] = l * (1.f / (float)B);
            part4[blockIdx.x * 4 + 1] = dz;
            part4[blockIdx.x * 4 + 2] = dzh;
        }
    }
    __syncthreads();
    // ---- backward of the 1-wide layer on the rows still in registers ----
    float g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = sdzh[wv * 4 + i];
    float* wr = wred + wv * 2 * K;
#pragma unroll
    for (int it = 0; it < KI; ++it) {
        const int k = lane * 4 + it * 256;
        if (k < K) {
            float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
            const float wk[4] = {c[it].x, c[it].y, c[it].z, c[it].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {               // rows in order: fixed summation order
                const float x[4] = {a[it][i].x, a[it][i].y, a[it][i].z, a[it][i].w};
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = g[i] * wk[e];
                    if (mul_act) v[e] *= rs_act_grad(mul_act, x[e], x[e], 0.f);
                    aw[e] = fmaf(g[i], x[e], aw[e]);
                    ab[e] += v[e];
                }
                const int m = r0 + wv * 4 + i;
                if (m < B) *reinterpret_cast<float4*>(dX + (long long)m * K + k) = make_float4(v[0], v[1], v[2], v[3]);
            }
            *reinterpret_cast<float4*>(wr + k) = make_float4(aw[0], aw[1], aw[2], aw[3]);
            *reinterpret_cast<float4*>(wr + K + k) = make_float4(ab[0], ab[1], ab[2], ab[3]);
        }
    }
    __syncthreads();
    float* prow = prow_all + (long long)blockIdx.x * 2 * K;
    for (int col = threadIdx.x; col < 2 * K; col += 256)
        prow[col] = ((wred[col] + wred[2 * K + col]) + wred[4 * K + col]) + wred[6 * K + col];      // wave order
}

// dst[b] = act(z[b])-style pointwise helpers on vectors -------------------------------------
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ z, float* __restrict__ h, long long n,
                                                      int act, const float* __restrict__ slope) {
    const float sl = slope ? slope[0] : 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) h[i] = rs_act(act, z[i], sl);
}

// dz = dh * act'(z);  for PReLU also the slope gradient partials: sum over z<=0 of dh*z
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ z,
                                                      const float* __restrict__ h, float* __restrict__ dz, long long n,
                                                      int act, const float* __restrict__ slope,
                                                      float* __restrict__ dslope_partials) {
    __shared__ float red[4];
    const float sl = slope ? slope[0] : 0.f;
    float ds = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float zz = z ? z[i] : h[i], hh = h ? h[i] : 0.f, g = dh[i];
        dz[i] = g * rs_act_grad(act, zz, hh, sl);
        if (act == RS_ACT_PRELU && zz <= 0.f) ds += g * zz;
    }
    if (dslope_partials) {
        const float tot = rs_block_sum<256>(ds, red);
        if (threadIdx.x == 0) dslope_partials[blockIdx.x] = tot;
    }
}

// ---- BatchNorm1d (training: batch statistics; eval: running statistics) ---------------------
// One workgroup per 32 features; 8 row-lanes x 32 feature-lanes; two passes (mean, then centred
// variance) for the accuracy of torch's batch_norm.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, int M, int N,
                                                       float* __restrict__ mean, float* __restrict__ var_biased) {
    __shared__ float part[8][33];
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    float s = 0.f;
    if (n < N) for (int m = r; m < M; m += 8) s += z[(long long)m * N + n];
    part[r][c] = s;
    __syncthreads();
    float mu = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mu += part[i][c];
    mu /= (float)M;
    __syncthreads();
    float q = 0.f;
    if (n < N) for (int m = r; m < M; m += 8) { const float dlt = z[(long long)m * N + n] - mu; q += dlt * dlt; }
    part[r][c] = q;
    __syncthreads();
    if (r == 0 && n < N) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) v += part[i][c];
        mean[n] = mu;
        var_biased[n] = v / (float)M;
    }
}

// h = act(gamma * (z - mean) * rsqrt(var + eps) + beta); optionally updates running stats
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, int M, int N,
                                                       const float* __restrict__ mean, const float* __restrict__ var,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int act, float* __restrict__ xhat,
                                                       float* __restrict__ h) {
    const long long total = (long long)M * N;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i % N);
        const float xh = (z[i] - mean[n]) * (1.f / sqrtf(var[n] + eps));
        if (xhat) xhat[i] = xh;
        h[i] = rs_act(act, gamma[n] * xh + beta[n], 0.f);
    }
}

__global__ void bn_running_kernel(const float* __restrict__ mean, const float* __restrict__ var_biased, int M, int N,
                                  float momentum, float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) {
        const float unb = M > 1 ? var_biased[n] * ((float)M / (float)(M - 1)) : var_biased[n];
        running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * mean[n];
        running_var[n] = (1.f - momentum) * running_var[n] + momentum * unb;
    }
}

// BatchNorm backward (training).  dy = dh * act'(...) is formed on the fly from h.
//   dbeta = sum dy, dgamma = sum dy*xhat, dz = gamma*rstd*(dy - dbeta/M - xhat*dgamma/M)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                            const float* __restrict__ xhat, int M, int N, int act,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float p1[8][33], p2[8][33];
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    float sb = 0.f, sg = 0.f;
    if (n < N)
        for (int m = r; m < M; m += 8) {
            const long long i = (long long)m * N + n;
            const float hh = h[i];
            const float dy = dh[i] * rs_act_grad(act, hh, hh, 0.f);   // relu/leaky: sign(h)=sign(z); tanh: from h
            sb += dy; sg += dy * xhat[i];
        }
    p1[r][c] = sb; p2[r][c] = sg;
    __syncthreads();
    if (r == 0 && n < N) {
        float b = 0.f, g = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { b += p1[i][c]; g += p2[i][c]; }
        dbeta[n] = b; dgamma[n] = g;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                           const float* __restrict__ xhat, int M, int N, int act,
                                                           const float* __restrict__ gamma, const float* __restrict__ var,
                                                           float eps, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, int m_total,
                                                           float* __restrict__ dz) {
    const long long total = (long long)M * N;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float invM = 1.f / (float)m_total;       // samples the statistics were taken over (all ranks' batches)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i % N);
        const float hh = h[i];
        const float dy = dh[i] * rs_act_grad(act, hh, hh, 0.f);
        const float rstd = 1.f / sqrtf(var[n] + eps);
        dz[i] = gamma[n] * rstd * (dy - dbeta[n] * invM - xhat[i] * dgamma[n] * invM);
    }
}

// ---- LayerNorm over the feature dimension: one wave per row ---------------------------------
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ z, int M, int N,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, int act, float* __restrict__ xhat,
                                                     float* __restrict__ rstd_out, float* __restrict__ h) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* zr = z + (long long)m * N;
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += zr[n];
    const float mu = rs_wave_sum(s) / (float)N;
    float q = 0.f;
    for (int n = lane; n < N; n += 64) { const float dlt = zr[n] - mu; q += dlt * dlt; }
    const float rstd = 1.f / sqrtf(rs_wave_sum(q) / (float)N + eps);
    if (lane == 0) rstd_out[m] = rstd;
    for (int n = lane; n < N; n += 64) {
        const float xh = (zr[n] - mu) * rstd;
        xhat[(long long)m * N + n] = xh;
        h[(long long)m * N + n] = rs_act(act, gamma[n] * xh + beta[n], 0.f);
    }
}

// dz = rstd * (g - mean(g) - xhat*mean(g*xhat)),  g = dy*gamma;  dy = dh*act'(h)
// also writes dy (into dy_out) so that dgamma/dbeta column sums can run as a second pass.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                     const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, int M, int N, int act,
                                                     float* __restrict__ dz, float* __restrict__ dy_out,
                                                     float* __restrict__ dyxhat_out) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const long long base = (long long)m * N;
    float s1 = 0.f, s2 = 0.f;
    for (int n = lane; n < N; n += 64) {
        const float hh = h[base + n];
        const float dy = dh[base + n] * rs_act_grad(act, hh, hh, 0.f);
        const float g = dy * gamma[n];
        s1 += g; s2 += g * xhat[base + n];
        dy_out[base + n] = dy;
        dyxhat_out[base + n] = dy * xhat[base + n];
    }
    s1 = rs_wave_sum(s1) / (float)N;
    s2 = rs_wave_sum(s2) / (float)N;
    const float r = rstd[m];
    for (int n = lane; n < N; n += 64) {
        const float g = dy_out[base + n] * gamma[n];
        dz[base + n] = r * (g - s1 - xhat[base + n] * s2);
    }
}

// ---- dropout (nn.Dropout in training mode, model/utils/Layers.py:33-37,46): y[i] = keep(i) ? x[i] / (1 - p) : 0 with
// the mask of philox.h ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ x, float* __restrict__ y, long long n,
                                                            DropKey key) {
    const DropConsts d = drop_consts(key);
    const long long nblk = (n + 3) / 4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        float f[4];
        drop_factors(d, b, f);
        const long long i0 = b * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long i = i0 + e;
            if (i < n) y[i] = f[e] != 0.f ? x[i] * f[e] : 0.f;
        }
    }
}

// y = a + b (residual add), flat
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = a[i] + b[i];
}

// y += a * x, flat
__global__ __launch_bounds__(256) void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y,
                                                   long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] += a * x[i];
}

// out[0] = base[0] + sum_i coef[i] * partial_sums_i  (adds the L2 penalty terms to the loss)
__global__ void loss_add_reg_kernel(float* loss, const float* __restrict__ partials, const int* __restrict__ seg_end,
                                    const float* __restrict__ coef, int nseg) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float l = loss[0];
    int beg = 0;
    for (int s = 0; s < nseg; ++s) {
        float t = 0.f;
        for (int i = beg; i < seg_end[s]; ++i) t += partials[i];
        l += coef[s] * t;
        beg = seg_end[s];
    }
    loss[0] = l;
}

int flat_blocks(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

// Loss head of the training step; any of logit_out/prob_out/loss_out/dz_out may be null.
// Labels come either as float (labels_f32) or int64 (labels_i64); exactly one must be non-null.
// sum_dz_out / hlast / act_last / dzh_out / sum_dzh_out: optional fused by-products (see kernel).
int rs_bce_head(const float* z0, const float* z1, const float* z2, const float* labels_f32,
                const int64_t* labels_i64, int B, float* logit_out, float* prob_out, float* loss_out,
                float* dz_out, float grad_scale, float* sum_dz_out, const float* hlast, int act_last, float* dzh_out,
                float* sum_dzh_out, hipStream_t stream) {
    RS_CHECK_ARG(B > 0 && z0, "rs_bce_head: empty batch or null logits");
    RS_CHECK_ARG((labels_f32 != nullptr) != (labels_i64 != nullptr), "rs_bce_head: pass exactly one label array");
    RS_CHECK_ARG(hlast || !(dzh_out || sum_dzh_out), "rs_bce_head: dzh outputs need hlast");
    RS_CHECK_ARG(act_last >= RS_ACT_NONE && act_last <= RS_ACT_LEAKYRELU, "rs_bce_head: bad act_last");
    hipLaunchKernelGGL(bce_head_kernel, dim3(1), dim3(1024), 0, stream, z0, z1, z2, labels_f32, labels_i64, B,
                       logit_out, prob_out, loss_out, dz_out, grad_scale, sum_dz_out, hlast, act_last, dzh_out,
                       sum_dzh_out);
    RS_CHECK_LAUNCH("rs_bce_head");
    return RS_OK;
}

// h[m] = act(X[m,:] . w + bias[0]): forward of a 1-wide linear layer (one wave per row).
int rs_rowdot_fwd(const float* X, int M, int K, const float* w, const float* bias, int act, float* h,
                  hipStream_t stream) {
    RS_CHECK_ARG(M >= 0 && K > 0, "rs_rowdot_fwd: bad sizes");
    if (M == 0) return RS_OK;
    RS_CHECK_ARG(X && w && h, "rs_rowdot_fwd: null argument");
    RS_CHECK_ARG(act >= RS_ACT_NONE && act <= RS_ACT_LEAKYRELU, "rs_rowdot_fwd: activation %d not fusable", act);
    RS_CHECK_ARG(((K & 3) != 0) || ((((uintptr_t)X | (uintptr_t)w) & 15) == 0), "rs_rowdot_fwd: operands must be 16-byte aligned");
    hipLaunchKernelGGL(rowdot_fwd_kernel, dim3(rs_cdiv(M, 4)), dim3(256), 0, stream, X, M, K, w, bias, act, h);
    RS_CHECK_LAUNCH("rs_rowdot_fwd");
    return RS_OK;
}

// dX[m,k] = dz[m] * w[k] * act'(X[m,k])  (mul_act = 0: no activation factor)
int rs_rowdot_bwd_dx(const float* dz, const float* w, const float* X, int mul_act, int M, int K, float* dX,
                     hipStream_t stream) {
    RS_CHECK_ARG(M >= 0 && K > 0, "rs_rowdot_bwd_dx: bad sizes");
    if (M == 0) return RS_OK;
    RS_CHECK_ARG(dz && w && dX && (X || !mul_act), "rs_rowdot_bwd_dx: null argument");
    hipLaunchKernelGGL(rowdot_bwd_dx_kernel, dim3(flat_blocks((long long)M * K)), dim3(256), 0, stream, dz, w, X, mul_act,
                       M, K, dX);
    RS_CHECK_LAUNCH("rs_rowdot_bwd_dx");
    return RS_OK;
}


// Scratch of the fused head entry points for a batch of M rows and a K-wide input: floats of `ws`.
size_t rs_head_workspace_floats(int M, int K) {
    const size_t g = (size_t)rs_cdiv(M, HEAD_ROWS);
    const size_t a = g * 4, b = g * 2 * (size_t)K;
    return a > b ? a : b;
}

// 1-wide last layer + sigmoid/BCE head in one launch (see head_fwd_bce_kernel).  z0, z1: the other logit
// parts (nullable; z1 needs z0).  sem: one zeroed int (left zero).  ws: rs_head_workspace_floats(B, K).
// sem == NULL: the three batch sums are left to the caller -- ws[g*4 + {0,1,2}], g < ceil(B/16), holds per 16 rows the
// loss / B, sum dz and sum dzh, to be added over g (rs_colsum_finish_multi with ld = 4); loss_out, sum_dz_out and
// sum_dzh_out are not written.
int rs_head_fwd_bce(const float* X, int B, int K, const float* w, const float* bias, int act, const float* z0,
                    const float* z1, const float* labels_f32, const int64_t* labels_i64, float grad_scale, float* h_out,
                    float* loss_out, float* dz_out, float* sum_dz_out, float* dzh_out, float* sum_dzh_out, float* ws,
                    int* sem, hipStream_t stream) {
    RS_CHECK_ARG(B > 0 && K > 0 && X && w && ws, "rs_head_fwd_bce: bad arguments");
    RS_CHECK_ARG((K & 3) == 0 && ((((uintptr_t)X | (uintptr_t)w) & 15) == 0), "rs_head_fwd_bce: K %% 4 and 16-byte alignment");
    RS_CHECK_ARG((labels_f32 != nullptr) != (labels_i64 != nullptr), "rs_head_fwd_bce: pass exactly one label array");
    RS_CHECK_ARG(z0 || !z1, "rs_head_fwd_bce: z1 without z0");
    RS_CHECK_ARG(act >= RS_ACT_NONE && act <= RS_ACT_LEAKYRELU, "rs_head_fwd_bce: activation %d not fusable", act);
    hipLaunchKernelGGL(head_fwd_bce_kernel, dim3(rs_cdiv(B, HEAD_ROWS)), dim3(256), 0, stream, X, B, K, w, bias, act, z0, z1,
                       labels_f32, labels_i64, grad_scale, h_out, loss_out, dz_out, sum_dz_out, dzh_out, sum_dzh_out, ws, sem);
    RS_CHECK_LAUNCH("rs_head_fwd_bce");
    return RS_OK;
}

// rs_head_fwd_bce (sem == NULL form: batch sums left in ws[4g + {0,1,2}]) and rs_head_bwd in one launch; X is read
// once.  part: [rs_head_bwd_rows(B), 2K] as for rs_head_bwd.  K % 4 == 0, K <= 1024.
int rs_head_fwd_bwd(const float* X, int B, int K, const float* w, const float* bias, int act, const float* z0,
                    const float* z1, const float* labels_f32, const int64_t* labels_i64, float grad_scale, float* h_out,
                    float* dz_out, float* dzh_out, float* ws, int mul_act, float* dX, float* part, hipStream_t stream) {
    RS_CHECK_ARG(B > 0 && K > 0 && X && w && ws && dX && part, "rs_head_fwd_bwd: bad arguments");
    RS_CHECK_ARG((K & 3) == 0 && K <= 1024 && ((((uintptr_t)X | (uintptr_t)w | (uintptr_t)dX) & 15) == 0),
                 "rs_head_fwd_bwd: K %% 4, K <= 1024 and 16-byte alignment");
    RS_CHECK_ARG((labels_f32 != nullptr) != (labels_i64 != nullptr), "rs_head_fwd_bwd: pass exactly one label array");
    RS_CHECK_ARG(z0 || !z1, "rs_head_fwd_bwd: z1 without z0");
    RS_CHECK_ARG(act >= RS_ACT_NONE && act <= RS_ACT_LEAKYRELU, "rs_head_fwd_bwd: activation %d not fusable", act);
    RS_CHECK_ARG(mul_act >= RS_ACT_NONE && mul_act <= RS_ACT_LEAKYRELU, "rs_head_fwd_bwd: activation %d not fusable", mul_act);
    const dim3 grid(rs_cdiv(B, HEAD_ROWS)), block(256);
    const size_t smem = (size_t)8 * K * sizeof(float);
    const int ki = rs_cdiv(K, 256);
#define RS_HEAD_FB(KI) hipLaunchKernelGGL(head_fwd_bwd_kernel<KI>, grid, block, smem, stream, X, B, K, w, bias, act, z0, z1, \
                                          labels_f32, labels_i64, grad_scale, h_out, dz_out, dzh_out, ws, mul_act, dX, part)
    if (ki == 1) RS_HEAD_FB(1); else if (ki == 2) RS_HEAD_FB(2); else if (ki == 3) RS_HEAD_FB(3); else RS_HEAD_FB(4);
#undef RS_HEAD_FB
    RS_CHECK_LAUNCH("rs_head_fwd_bwd");
    return RS_OK;
}

// Backward of the 1-wide layer: dX [B,K] and part [rs_head_bwd_rows(B), 2K], the per-16-row partial sums of
// [dW | dbp] (see head_bwd_kernel); their sums over rows are dW and dbp (rs_colsum_finish_multi).
int rs_head_bwd_rows(int B) { return rs_cdiv(B, HEAD_ROWS); }
int rs_head_bwd(const float* dzh, const float* w, const float* X, int mul_act, int B, int K, float* dX, float* part,
                hipStream_t stream) {
    RS_CHECK_ARG(B > 0 && K > 0 && K <= 256 * HEAD_KPT, "rs_head_bwd: bad sizes (K <= %d)", 256 * HEAD_KPT);
    RS_CHECK_ARG(dzh && w && X && dX && part, "rs_head_bwd: null argument");
    hipLaunchKernelGGL(head_bwd_kernel, dim3(rs_cdiv(B, HEAD_ROWS)), dim3(256), 0, stream, dzh, w, X, mul_act, B, K, dX, part);
    RS_CHECK_LAUNCH("rs_head_bwd");
    return RS_OK;
}

// The 1-wide layer over cat([X1, X2], 1) without the concatenation (DCN.py:31,52: Linear(D + h_last, 1) over the cross
// network's output and the deep tower's): w [K1 + K2]; backward as rs_head_bwd with dX and the columns of part split
// the same way, mul_act2 = the activation whose output X2 is (0: none).  K1 + K2 <= 1024.
int rs_rowdot2_fwd(const float* X1, int K1, const float* X2, int K2, int M, const float* w, const float* bias, int act,
                   float* h, hipStream_t stream) {
    RS_CHECK_ARG(M >= 0 && K1 > 0 && K2 > 0, "rs_rowdot2_fwd: bad sizes");
    if (M == 0) return RS_OK;
    RS_CHECK_ARG(X1 && X2 && w && h, "rs_rowdot2_fwd: null argument");
    RS_CHECK_ARG(((K1 | K2) & 3) != 0 || ((reinterpret_cast<uintptr_t>(X1) | reinterpret_cast<uintptr_t>(X2) |
                                           reinterpret_cast<uintptr_t>(w)) & 15) == 0,
                 "rs_rowdot2_fwd: operands of widths that are multiples of four must be 16-byte aligned");
    hipLaunchKernelGGL(rowdot2_fwd_kernel, dim3(rs_cdiv(M, 4)), dim3(256), 0, stream, X1, K1, X2, K2, M, w, bias, act, h);
    RS_CHECK_LAUNCH("rs_rowdot2_fwd");
    return RS_OK;
}

int rs_head2_bwd(const float* dzh, const float* w, const float* X1, int K1, const float* X2, int K2, int mul_act2, int B,
                 float* dX1, float* dX2, float* part, hipStream_t stream) {
    RS_CHECK_ARG(B > 0 && K1 > 0 && K2 > 0 && K1 + K2 <= 256 * HEAD_KPT, "rs_head2_bwd: bad sizes (K1 + K2 <= %d)", 256 * HEAD_KPT);
    RS_CHECK_ARG(dzh && w && X1 && X2 && dX1 && dX2 && part, "rs_head2_bwd: null argument");
    hipLaunchKernelGGL(head2_bwd_kernel, dim3(rs_cdiv(B, HEAD_ROWS)), dim3(256), 0, stream, dzh, w, X1, K1, X2, K2, mul_act2,
                       B, dX1, dX2, part);
    RS_CHECK_LAUNCH("rs_head2_bwd");
    return RS_OK;
}

int rs_act_fwd(const float* z, float* h, long long n, int act, const float* slope, hipStream_t stream) {
    RS_CHECK_ARG(n >= 0 && (n == 0 || (z && h)), "rs_act_fwd: bad arguments");
    RS_CHECK_ARG(act != RS_ACT_PRELU || slope, "rs_act_fwd: PReLU needs its slope");
    if (n == 0) return RS_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(flat_blocks(n)), dim3(256), 0, stream, z, h, n, act, slope);
    RS_CHECK_LAUNCH("rs_act_fwd");
    return RS_OK;
}

int rs_act_bwd_num_partials(long long n) { return flat_blocks(n); }

// dz = dh * act'(z).  z may be null for activations whose derivative follows from h.
// dslope_partials (PReLU only): rs_act_bwd_num_partials(n) floats, summed by the caller.
int rs_act_bwd(const float* dh, const float* z, const float* h, float* dz, long long n, int act, const float* slope,
               float* dslope_partials, hipStream_t stream) {
    RS_CHECK_ARG(n >= 0 && (n == 0 || (dh && dz && (z || h))), "rs_act_bwd: bad arguments");
    RS_CHECK_ARG(act != RS_ACT_PRELU || (slope && z), "rs_act_bwd: PReLU needs slope and z");
    if (n == 0) return RS_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(flat_blocks(n)), dim3(256), 0, stream, dh, z, h, dz, n, act, slope,
                       dslope_partials);
    RS_CHECK_LAUNCH("rs_act_bwd");
    return RS_OK;
}

// BatchNorm1d forward over z [M, N] (model/utils/Layers.py:27,44 in train mode when training != 0).
// training: computes batch mean / biased var into mean,var and updates running stats (momentum 0.1
// semantics of torch) when running_* are given; eval: uses running stats passed as mean,var.
int rs_bn_fwd(const float* z, int M, int N, const float* gamma, const float* beta, float eps, int act, int training,
              float momentum, float* mean, float* var, float* running_mean, float* running_var, float* xhat,
              float* h, hipStream_t stream) {
    RS_CHECK_ARG(M > 0 && N > 0 && z && gamma && beta && mean && var && h, "rs_bn_fwd: bad arguments");
    RS_CHECK_ARG(act >= RS_ACT_NONE && act <= RS_ACT_LEAKYRELU, "rs_bn_fwd: activation %d not fusable", act);
    if (training) {
        hipLaunchKernelGGL(bn_stats_kernel, dim3(rs_cdiv(N, 32)), dim3(256), 0, stream, z, M, N, mean, var);
        if (running_mean && running_var)
            hipLaunchKernelGGL(bn_running_kernel, dim3(rs_cdiv(N, 256)), dim3(256), 0, stream, mean, var, M, N, momentum,
                               running_mean, running_var);
    }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(flat_blocks((long long)M * N)), dim3(256), 0, stream, z, M, N, mean, var,
                       gamma, beta, eps, act, xhat, h);
    RS_CHECK_LAUNCH("rs_bn_fwd");
    return RS_OK;
}

int rs_bn_bwd(const float* dh, const float* h, const float* xhat, int M, int N, int act, const float* gamma,
              const float* var, float eps, float* dgamma, float* dbeta, float* dz, hipStream_t stream) {
    RS_CHECK_ARG(M > 0 && N > 0 && dh && h && xhat && gamma && var && dgamma && dbeta && dz, "rs_bn_bwd: bad arguments");
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(rs_cdiv(N, 32)), dim3(256), 0, stream, dh, h, xhat, M, N, act, dgamma,
                       dbeta);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(flat_blocks((long long)M * N)), dim3(256), 0, stream, dh, h, xhat, M, N,
                       act, gamma, var, eps, dgamma, dbeta, M, dz);
    RS_CHECK_LAUNCH("rs_bn_bwd");
    return RS_OK;
}

// The pieces of rs_bn_fwd / rs_bn_bwd for a batch that is spread over several GPUs (one process each): the caller
// combines the per-rank results between them, so that mean, variance and the two backward sums are those of the union
// batch -- what the reference's single-process BatchNorm1d sees (Layers.py:27,44).
//   rs_bn_stats       mean and biased variance (centred at that mean) of this rank's M rows
//   rs_bn_bwd_reduce  this rank's dbeta = sum dy and dgamma = sum dy * xhat (its share of the parameter gradients)
//   rs_bn_bwd_apply   dz from sums over all m_total rows of the union batch
// The forward's normalisation itself is rs_bn_fwd with training = 0 and the combined statistics as mean / var.
int rs_bn_stats(const float* z, int M, int N, float* mean, float* var, hipStream_t stream) {
    RS_CHECK_ARG(M > 0 && N > 0 && z && mean && var, "rs_bn_stats: bad arguments");
    hipLaunchKernelGGL(bn_stats_kernel, dim3(rs_cdiv(N, 32)), dim3(256), 0, stream, z, M, N, mean, var);
    RS_CHECK_LAUNCH("rs_bn_stats");
    return RS_OK;
}

int rs_bn_bwd_reduce(const float* dh, const float* h, const float* xhat, int M, int N, int act, float* dgamma,
                     float* dbeta, hipStream_t stream) {
    RS_CHECK_ARG(M > 0 && N > 0 && dh && h && xhat && dgamma && dbeta, "rs_bn_bwd_reduce: bad arguments");
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(rs_cdiv(N, 32)), dim3(256), 0, stream, dh, h, xhat, M, N, act, dgamma,
                       dbeta);
    RS_CHECK_LAUNCH("rs_bn_bwd_reduce");
    return RS_OK;
}

int rs_bn_bwd_apply(const float* dh, const float* h, const float* xhat, int M, int N, int act, const float* gamma,
                    const float* var, float eps, const float* dgamma_all, const float* dbeta_all, int m_total, float* dz,
                    hipStream_t stream) {
    RS_CHECK_ARG(M > 0 && N > 0 && m_total >= M && dh && h && xhat && gamma && var && dgamma_all && dbeta_all && dz,
                 "rs_bn_bwd_apply: bad arguments");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(flat_blocks((long long)M * N)), dim3(256), 0, stream, dh, h, xhat, M, N,
                       act, gamma, var, eps, dgamma_all, dbeta_all, m_total, dz);
    RS_CHECK_LAUNCH("rs_bn_bwd_apply");
    return RS_OK;
}

int rs_ln_fwd(const float* z, int M, int N, const float* gamma, const float* beta, float eps, int act, float* xhat,
              float* rstd, float* h, hipStream_t stream) {
    RS_CHECK_ARG(M > 0 && N > 0 && z && gamma && beta && xhat && rstd && h, "rs_ln_fwd: bad arguments");
    RS_CHECK_ARG(act >= RS_ACT_NONE && act <= RS_ACT_LEAKYRELU, "rs_ln_fwd: activation %d not fusable", act);
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(rs_cdiv(M, 4)), dim3(256), 0, stream, z, M, N, gamma, beta, eps, act, xhat,
                       rstd, h);
    RS_CHECK_LAUNCH("rs_ln_fwd");
    return RS_OK;
}

// LayerNorm backward: dz, plus dy and dy*xhat (both [M, N]) whose column sums are dbeta / dgamma.
int rs_ln_bwd(const float* dh, const float* h, const float* xhat, const float* rstd, const float* gamma, int M, int N,
              int act, float* dz, float* dy, float* dyxhat, hipStream_t stream) {
    RS_CHECK_ARG(M > 0 && N > 0 && dh && h && xhat && rstd && gamma && dz && dy && dyxhat, "rs_ln_bwd: bad arguments");
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(rs_cdiv(M, 4)), dim3(256), 0, stream, dh, h, xhat, rstd, gamma, M, N, act, dz,
                       dy, dyxhat);
    RS_CHECK_LAUNCH("rs_ln_bwd");
    return RS_OK;
}

// y += a * x  (L2 regularisation gradient lambda * w, model/utils/BaseModel.py:214-231)
int rs_axpy_f32(float a, const float* x, float* y, long long n, hipStream_t stream) {
    RS_CHECK_ARG(n >= 0 && (n == 0 || (x && y)), "rs_axpy_f32: bad arguments");
    if (n == 0) return RS_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(flat_blocks(n)), dim3(256), 0, stream, a, x, y, n);
    RS_CHECK_LAUNCH("rs_axpy_f32");
    return RS_OK;
}

// loss[0] += sum_s coef[s] * sum(partials[seg_end[s-1] : seg_end[s]])   (single thread, fixed order)
int rs_loss_add_reg(float* loss, const float* partials, const int* seg_end, const float* coef, int nseg,
                    hipStream_t stream) {
    RS_CHECK_ARG(loss && partials && seg_end && coef && nseg >= 0, "rs_loss_add_reg: bad arguments");
    hipLaunchKernelGGL(loss_add_reg_kernel, dim3(1), dim3(64), 0, stream, loss, partials, seg_end, coef, nseg);
    RS_CHECK_LAUNCH("rs_loss_add_reg");
    return RS_OK;
}

// y = dropout_mask(seed, salt, call) * x / (1 - p); the same call on a gradient is the backward pass.
// call = call_host + *call_dev (call_dev may be null): a device-side step counter keeps a replayed HIP graph drawing
// fresh masks.  x == y (in place) is allowed.
int rs_dropout_apply(const float* x, float* y, long long n, float p, unsigned long long seed, unsigned salt,
                     unsigned long long call_host, const long long* call_dev, hipStream_t stream) {
    RS_CHECK_ARG(n >= 0 && (n == 0 || (x && y)), "rs_dropout_apply: bad arguments");
    RS_CHECK_ARG(p >= 0.f && p <= 1.f, "rs_dropout_apply: dropout probability has to be between 0 and 1, but got %f", (double)p);
    if (n == 0) return RS_OK;
    const DropKey key{p, salt, seed, call_host, call_dev};
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(flat_blocks((n + 3) / 4)), dim3(256), 0, stream, x, y, n, key);
    RS_CHECK_LAUNCH("rs_dropout_apply");
    return RS_OK;
}

int rs_add_f32(const float* a, const float* b, float* y, long long n, hipStream_t stream) {
    RS_CHECK_ARG(n >= 0 && (n == 0 || (a && b && y)), "rs_add_f32: bad arguments");
    if (n == 0) return RS_OK;
    hipLaunchKernelGGL(add_kernel, dim3(flat_blocks(n)), dim3(256), 0, stream, a, b, y, n);
    RS_CHECK_LAUNCH("rs_add_f32");
    return RS_OK;
}

}  // extern "C"
