// K2: feature-interaction kernels on already-gathered embeddings emb[B, F, d] (fp32).
#include "common.h"
#include "plan.h"

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


// ---- pair indexing shared by the PNN / AFM kernels -----------------------------------------------
// Pairs (i, j), i < j, in row-major order (model/PNN/PNN.py:69-70, model/AFM/AFM.py:51-52):
// p = i*(2F-i-1)/2 + (j-i-1).
__device__ __forceinline__ int pair_start(int i, int F) { return i * (2 * F - i - 1) / 2; }
__device__ __forceinline__ int pair_index(int i, int j, int F) { return pair_start(i, F) + (j - i - 1); }
__device__ __forceinline__ int pair_index_any(int a, int b, int F) { return a < b ? pair_index(a, b, F) : pair_index(b, a, F); }
__device__ __forceinline__ void pair_decode(int p, int F, int* io, int* jo) {
    const float t = (float)(2 * F - 1);
    int i = (int)((t - sqrtf(fmaxf(t * t - 8.f * (float)p, 0.f))) * 0.5f);
    i = i < 0 ? 0 : (i > F - 2 ? F - 2 : i);
    while (i < F - 2 && pair_start(i + 1, F) <= p) ++i;
    while (i > 0 && pair_start(i, F) > p) --i;
    *io = i;
    *jo = p - pair_start(i, F) + i + 1;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- PNN inner product (model/PNN/PNN.py:65-77) ---------------------------------------------------
// One wave per sample: the sample's F*d embedding values sit in LDS, lane l computes pairs l, l+64, ...
// and the (B, P) output row is written in pair order (coalesced).
__global__ __launch_bounds__(256) void pnn_inner_fwd_kernel(const float* __restrict__ emb, int B, int F, int d,
                                                            float* __restrict__ out) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    const bool valid = b < B;
    const int fd = F * d, P = F * (F - 1) / 2;
    float* e = sm + w * fd;
    if (valid) {
        const float* src = emb + (long long)b * fd;
        for (int i = lane; i < fd; i += 64) e[i] = src[i];
    }
    __syncthreads();
    if (!valid) return;
    for (int p = lane; p < P; p += 64) {
        int i, j;
        pair_decode(p, F, &i, &j);
        float acc = 0.f;
        for (int k = 0; k < d; ++k) acc = fmaf(e[i * d + k], e[j * d + k], acc);
        out[(long long)b * P + p] = acc;
    }
}

// demb[b, f, k] = sum_{j != f} dP[b, pair(f, j)] * e[b, j, k]
__global__ __launch_bounds__(256) void pnn_inner_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ dout,
                                                            int B, int F, int d, float* __restrict__ demb) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    const bool valid = b < B;
    const int fd = F * d, P = F * (F - 1) / 2;
    float* e = sm + w * (fd + P);
    float* g = e + fd;
    if (valid) {
        const float* src = emb + (long long)b * fd;
        for (int i = lane; i < fd; i += 64) e[i] = src[i];
        const float* gs = dout + (long long)b * P;
        for (int i = lane; i < P; i += 64) g[i] = gs[i];
    }
    __syncthreads();
    if (!valid) return;
    for (int idx = lane; idx < fd; idx += 64) {
        const int f = idx / d, k = idx - f * d;
        float acc = 0.f;
        for (int j = 0; j < F; ++j) {
            if (j == f) continue;
            acc = fmaf(g[pair_index_any(f, j, F)], e[j * d + k], acc);
        }
        demb[(long long)b * fd + idx] = acc;
    }
}

// ---- PNN outer product (model/PNN/PNN.py:80-103) --------------------------------------------------
// out[b, p, a, c] = e_i[a] * e_j[c] for pair p = (i, j).  One workgroup per sample, the (i, j) table of
// all pairs in LDS; the kernel is bound by the HBM write of P*d*d floats per sample.
template <int VEC>
__global__ __launch_bounds__(256) void pnn_outer_fwd_kernel(const float* __restrict__ emb, int B, int F, int d,
                                                            float* __restrict__ out) {
    extern __shared__ float sm[];
    const int fd = F * d, P = F * (F - 1) / 2, dd = d * d;
    float* e = sm;
    int* tab = reinterpret_cast<int*>(sm + fd);
    const int b = blockIdx.x;
    const float* src = emb + (long long)b * fd;
    for (int i = threadIdx.x; i < fd; i += 256) e[i] = src[i];
    for (int p = threadIdx.x; p < P; p += 256) {
        int i, j;
        pair_decode(p, F, &i, &j);
        tab[p] = i | (j << 16);
    }
    __syncthreads();
    float* o = out + (long long)b * P * dd;
    const int total = P * dd / VEC;
    for (int q = threadIdx.x; q < total; q += 256) {
        const int x = q * VEC;
        const int p = x / dd, r = x - p * dd;
        const int a = r / d, c = r - a * d;
        const int ij = tab[p], i = ij & 0xffff, j = ij >> 16;
        const float ea = e[i * d + a];
        if constexpr (VEC == 4) {
            const float4 ej = *reinterpret_cast<const float4*>(e + j * d + c);
            *reinterpret_cast<float4*>(o + x) = make_float4(ea * ej.x, ea * ej.y, ea * ej.z, ea * ej.w);
        } else {
            o[x] = ea * e[j * d + c];
        }
    }
}

// demb[b,f,k] = sum_{j>f} sum_c dO[p(f,j), k, c] e_j[c] + sum_{i<f} sum_a dO[p(i,f), a, k] e_i[a]
__global__ __launch_bounds__(256) void pnn_outer_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ dout,
                                                            int B, int F, int d, float* __restrict__ demb) {
    extern __shared__ float sm[];
    const int fd = F * d, P = F * (F - 1) / 2, dd = d * d;
    float* e = sm;
    const int b = blockIdx.x;
    const float* src = emb + (long long)b * fd;
    for (int i = threadIdx.x; i < fd; i += 256) e[i] = src[i];
    __syncthreads();
    const float* g = dout + (long long)b * P * dd;
    for (int idx = threadIdx.x; idx < fd; idx += 256) {
        const int f = idx / d, k = idx - f * d;
        float acc = 0.f;
        for (int j = f + 1; j < F; ++j) {
            const float* row = g + (long long)pair_index(f, j, F) * dd + k * d;
            for (int c = 0; c < d; ++c) acc = fmaf(row[c], e[j * d + c], acc);
        }
        for (int i = 0; i < f; ++i) {
            const float* col = g + (long long)pair_index(i, f, F) * dd + k;
            for (int a = 0; a < d; ++a) acc = fmaf(col[a * d], e[i * d + a], acc);
        }
        demb[(long long)b * fd + idx] = acc;
    }
}

// ---- DCN cross network (model/DCN/DCN.py:48-52 in its intended (B, D) shape) -----------------------
// x_{l+1} = x0 * (x_l . w_l) + b_l + x_l.  One wave per sample; element n lives in lane n % 64 throughout,
// so the running x needs no cross-lane traffic besides the dot-product reduction.  S[b, l] = x_l . w_l is
// kept for the backward pass.
__global__ __launch_bounds__(256) void dcn_cross_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ w,
                                                            const float* __restrict__ bias, int B, int D, int L,
                                                            float* __restrict__ out, float* __restrict__ S) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    if (b >= B) return;                     // no workgroup barrier below: every LDS word is lane-private
    float* x = sm + wv * D;
    const float* xs = x0 + (long long)b * D;
    for (int n = lane; n < D; n += 64) x[n] = xs[n];
    for (int l = 0; l < L; ++l) {
        float part = 0.f;
        for (int n = lane; n < D; n += 64) part = fmaf(x[n], w[l * D + n], part);
        const float s = rs_wave_sum(part);
        if (lane == 0 && S) S[(long long)b * L + l] = s;
        for (int n = lane; n < D; n += 64) x[n] = xs[n] * s + bias[l * D + n] + x[n];
    }
    for (int n = lane; n < D; n += 64) out[(long long)b * D + n] = x[n];
}

// Backward of the cross network.  One wave per workgroup walks samples b, b+G, ...; parameter gradients
// accumulate in lane-private LDS words and leave as one partial row per workgroup:
// partials[g] = [dW (L*D) | dB (L*D)], summed over g by the caller (rs_colsum_f32) in fixed order.
__global__ __launch_bounds__(64) void dcn_cross_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ S,
                                                           const float* __restrict__ dout, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int B, int D, int L,
                                                           float* __restrict__ dx0, float* __restrict__ partials) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x;
    float* xl = sm;                   // [L][D]  x_l, l = 0..L-1
    float* accW = xl + L * D;         // [L][D]
    float* accB = accW + L * D;       // [L][D]
    float* dx = accB + L * D;         // [D]
    float* d0 = dx + D;               // [D]
    for (int i = lane; i < 2 * L * D; i += 64) accW[i] = 0.f;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const float* xs = x0 + (long long)b * D;
        const float* sb = S + (long long)b * L;
        for (int n = lane; n < D; n += 64) {
            float x = xs[n];
            const float x0n = x;
            for (int l = 0; l < L; ++l) {
                xl[l * D + n] = x;
                x = x0n * sb[l] + bias[l * D + n] + x;
            }
            dx[n] = dout[(long long)b * D + n];
            d0[n] = 0.f;
        }
        for (int l = L - 1; l >= 0; --l) {
            float part = 0.f;
            for (int n = lane; n < D; n += 64) part = fmaf(dx[n], xs[n], part);
            const float ds = rs_wave_sum(part);
            const float s = sb[l];
            for (int n = lane; n < D; n += 64) {
                const float g = dx[n];
                accB[l * D + n] += g;
                accW[l * D + n] = fmaf(ds, xl[l * D + n], accW[l * D + n]);
                d0[n] = fmaf(g, s, d0[n]);
                dx[n] = fmaf(ds, w[l * D + n], g);
            }
        }
        for (int n = lane; n < D; n += 64) dx0[(long long)b * D + n] = d0[n] + dx[n];
    }
    float* prow = partials + (long long)blockIdx.x * 2 * L * D;
    for (int i = lane; i < 2 * L * D; i += 64) prow[i] = accW[i];
}

// ---- NFM bi-interaction in the reference's form (model/NFM/NFM.py:42-44) ---------------------------
// out[b,c] = 0.5 * ((sum_n e_n v_nc)^2 - sum_n e_n^2 v_nc^2), e = flattened (B, D) embeddings, v (D, K).
constexpr int NFM_KMAX = 16;
__global__ __launch_bounds__(256) void nfm_bi_fwd_kernel(const float* __restrict__ E, const float* __restrict__ V, int B,
                                                         int D, int K, float* __restrict__ out, float* __restrict__ s_out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float s[NFM_KMAX], q[NFM_KMAX];
#pragma unroll
    for (int c = 0; c < NFM_KMAX; ++c) s[c] = q[c] = 0.f;
    for (int n = lane; n < D; n += 64) {
        const float e = E[(long long)b * D + n];
#pragma unroll
        for (int c = 0; c < NFM_KMAX; ++c) {
            if (c < K) {
                const float v = V[n * K + c];
                s[c] = fmaf(e, v, s[c]);
                q[c] = fmaf(e * e, v * v, q[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NFM_KMAX; ++c) {
        if (c < K) {
            const float sc = rs_wave_sum(s[c]), qc = rs_wave_sum(q[c]);
            if (lane == 0) {
                out[(long long)b * K + c] = 0.5f * (sc * sc - qc);
                if (s_out) s_out[(long long)b * K + c] = sc;
            }
        }
    }
}

// dE[b,n] = sum_c g_bc (s_bc v_nc - e_bn v_nc^2);  dV[n,c] = sum_b g_bc e_bn (s_bc - e_bn v_nc)
// (dV as per-workgroup partial rows, like the cross network's parameter gradients).
__global__ __launch_bounds__(64) void nfm_bi_bwd_kernel(const float* __restrict__ E, const float* __restrict__ V,
                                                        const float* __restrict__ s_in, const float* __restrict__ dout,
                                                        int B, int D, int K, float* __restrict__ dE,
                                                        float* __restrict__ partials) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x;
    float* acc = sm;                  // [D][K], row n private to lane n % 64
    for (int i = lane; i < D; i += 64)
        for (int c = 0; c < K; ++c) acc[i * K + c] = 0.f;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        float g[NFM_KMAX], s[NFM_KMAX];
#pragma unroll
        for (int c = 0; c < NFM_KMAX; ++c) {
            g[c] = c < K ? dout[(long long)b * K + c] : 0.f;
            s[c] = c < K ? s_in[(long long)b * K + c] : 0.f;
        }
        for (int n = lane; n < D; n += 64) {
            const float e = E[(long long)b * D + n];
            float de = 0.f;
#pragma unroll
            for (int c = 0; c < NFM_KMAX; ++c) {
                if (c < K) {
                    const float v = V[n * K + c];
                    de = fmaf(g[c], s[c] * v - e * v * v, de);
                    acc[n * K + c] = fmaf(g[c] * e, s[c] - e * v, acc[n * K + c]);
                }
            }
            dE[(long long)b * D + n] = de;
        }
    }
    float* prow = partials + (long long)blockIdx.x * D * K;
    for (int i = lane; i < D; i += 64)
        for (int c = 0; c < K; ++c) prow[i * K + c] = acc[i * K + c];
}

// ---- AFM attention pooling (model/AFM/AFM.py:19-24, 36-42, 47-56) ----------------------------------
// Per sample: p_ij = e_i (.) e_j for the P pairs, score s_ij = h . relu(W p_ij + b), a = softmax over the
// pair axis, pooled = sum a_ij p_ij, out = pooled . pw + pb.  Nothing of size P*d ever leaves the chip:
// a wave keeps the sample's embeddings and the P scores in LDS and recomputes p_ij where it needs it.
template <int D>
__device__ __forceinline__ float afm_score(const float* pk, const float* sW, const float* sb, const float* sh, int A) {
    float s = 0.f;
    for (int a = 0; a < A; ++a) {
        float z = sb[a];
#pragma unroll
        for (int k = 0; k < D; ++k) z = fmaf(sW[a * D + k], pk[k], z);
        s = fmaf(sh[a], fmaxf(z, 0.f), s);
    }
    return s;
}

template <int D>
__global__ __launch_bounds__(64) void afm_fwd_kernel(const float* __restrict__ emb, int B, int F, int A,
                                                     const float* __restrict__ W, const float* __restrict__ bA,
                                                     const float* __restrict__ h, const float* __restrict__ pw,
                                                     const float* __restrict__ pb, float* __restrict__ out) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x;
    const int fd = F * D, P = F * (F - 1) / 2;
    float* sW = sm;                 // [A*D]
    float* sb = sW + A * D;         // [A]
    float* sh = sb + A;             // [A]
    float* spw = sh + A;            // [D]
    float* e = spw + D;             // [F*D]
    float* sc = e + fd;             // [P]
    for (int i = lane; i < A * D; i += 64) sW[i] = W[i];
    for (int i = lane; i < A; i += 64) { sb[i] = bA[i]; sh[i] = h[i]; }
    for (int i = lane; i < D; i += 64) spw[i] = pw[i];
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        const float* src = emb + (long long)b * fd;
        for (int i = lane; i < fd; i += 64) e[i] = src[i];
        __syncthreads();
        float m = -INFINITY;
        for (int p = lane; p < P; p += 64) {
            int i, j;
            pair_decode(p, F, &i, &j);
            float pk[D];
#pragma unroll
            for (int k = 0; k < D; ++k) pk[k] = e[i * D + k] * e[j * D + k];
            const float s = afm_score<D>(pk, sW, sb, sh, A);
            sc[p] = s;
            m = fmaxf(m, s);
        }
        m = wave_max(m);
        float sum = 0.f, pooled[D];
#pragma unroll
        for (int k = 0; k < D; ++k) pooled[k] = 0.f;
        for (int p = lane; p < P; p += 64) {
            int i, j;
            pair_decode(p, F, &i, &j);
            const float ex = expf(sc[p] - m);
            sum += ex;
#pragma unroll
            for (int k = 0; k < D; ++k) pooled[k] = fmaf(ex, e[i * D + k] * e[j * D + k], pooled[k]);
        }
        sum = rs_wave_sum(sum);
        float o = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) o = fmaf(pooled[k], spw[k], o);
        o = rs_wave_sum(o) / sum;
        if (lane == 0) out[b] = o + pb[0];
    }
}

// Backward.  Pair-level quantities (attention a_p, score gradient ds_p) are rebuilt in LDS; then for every
// field i the wave's lanes take the partner fields j, form dL/dp_ij and reduce dL/de_i over the lanes
// through an LDS transpose in fixed order (no atomics).  Parameter gradients accumulate in registers over
// the samples of the workgroup and leave as one partial row:
// [dW (A*D) | db (A) | dh (A) | dpw (D) | dpb (1)].
template <int D, int A>
__global__ __launch_bounds__(64) void afm_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ dout, int B,
                                                     int F, const float* __restrict__ W, const float* __restrict__ bA,
                                                     const float* __restrict__ h, const float* __restrict__ pw,
                                                     float* __restrict__ demb, float* __restrict__ partials) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x;
    const int fd = F * D, P = F * (F - 1) / 2;
    float* sW = sm;                 // [A*D]
    float* sb = sW + A * D;         // [A]
    float* sh = sb + A;             // [A]
    float* spw = sh + A;            // [D]
    float* e = spw + D;             // [F*D]
    float* sc = e + fd;             // [P]  attention a_p
    float* sds = sc + P;            // [P]  dL/ds_p
    float* ct = sds + P;            // [64*(D+1)]
    for (int i = lane; i < A * D; i += 64) sW[i] = W[i];
    for (int i = lane; i < A; i += 64) { sb[i] = bA[i]; sh[i] = h[i]; }
    for (int i = lane; i < D; i += 64) spw[i] = pw[i];
    float accW[A * D], accb[A], acch[A], accpw[D], accpb = 0.f;
#pragma unroll
    for (int i = 0; i < A * D; ++i) accW[i] = 0.f;
#pragma unroll
    for (int i = 0; i < A; ++i) accb[i] = acch[i] = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) accpw[i] = 0.f;

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        const float* src = emb + (long long)b * fd;
        for (int i = lane; i < fd; i += 64) e[i] = src[i];
        __syncthreads();
        const float go = dout[b];
        float m = -INFINITY;
        for (int p = lane; p < P; p += 64) {
            int i, j;
            pair_decode(p, F, &i, &j);
            float pk[D];
#pragma unroll
            for (int k = 0; k < D; ++k) pk[k] = e[i * D + k] * e[j * D + k];
            const float s = afm_score<D>(pk, sW, sb, sh, A);
            sc[p] = s;
            m = fmaxf(m, s);
        }
        m = wave_max(m);
        float sum = 0.f;
        for (int p = lane; p < P; p += 64) {
            const float ex = expf(sc[p] - m);
            sc[p] = ex;
            sum += ex;
        }
        const float inv = 1.f / rs_wave_sum(sum);
        float gbar = 0.f, pooled[D];
#pragma unroll
        for (int k = 0; k < D; ++k) pooled[k] = 0.f;
        for (int p = lane; p < P; p += 64) {
            int i, j;
            pair_decode(p, F, &i, &j);
            const float a = sc[p] * inv;
            sc[p] = a;
            float g = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const float pk = e[i * D + k] * e[j * D + k];
                g = fmaf(spw[k], pk, g);
                pooled[k] = fmaf(a, pk, pooled[k]);
            }
            g *= go;
            sds[p] = g;
            gbar = fmaf(a, g, gbar);
        }
        gbar = rs_wave_sum(gbar);
#pragma unroll
        for (int k = 0; k < D; ++k) accpw[k] = fmaf(go, pooled[k], accpw[k]);   // lane partials; summed at the end
        if (lane == 0) accpb += go;
        for (int p = lane; p < P; p += 64) sds[p] = sc[p] * (sds[p] - gbar);
        __syncthreads();

        for (int i = 0; i < F; ++i) {
            float tot = 0.f;
            for (int j0 = 0; j0 < F; j0 += 64) {
                const int j = j0 + lane;
                float c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) c[k] = 0.f;
                if (j < F && j != i) {
                    const int p = pair_index_any(i, j, F);
                    const float a = sc[p], ds = sds[p];
                    float pk[D], dp[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        pk[k] = e[i * D + k] * e[j * D + k];
                        dp[k] = a * go * spw[k];
                    }
#pragma unroll
                    for (int a_ = 0; a_ < A; ++a_) {
                        float z = sb[a_];
#pragma unroll
                        for (int k = 0; k < D; ++k) z = fmaf(sW[a_ * D + k], pk[k], z);
                        if (z > 0.f) {
                            const float dz = ds * sh[a_];
#pragma unroll
                            for (int k = 0; k < D; ++k) dp[k] = fmaf(dz, sW[a_ * D + k], dp[k]);
                            if (i < j) {             // every pair contributes to the parameters once
#pragma unroll
                                for (int k = 0; k < D; ++k) accW[a_ * D + k] = fmaf(dz, pk[k], accW[a_ * D + k]);
                                accb[a_] += dz;
                                acch[a_] = fmaf(ds, z, acch[a_]);
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < D; ++k) c[k] = dp[k] * e[j * D + k];
                }
#pragma unroll
                for (int k = 0; k < D; ++k) ct[lane * (D + 1) + k] = c[k];
                __syncthreads();
                if (lane < D) {
                    float t = 0.f;
                    for (int l = 0; l < 64; ++l) t += ct[l * (D + 1) + lane];
                    tot += t;
                }
                __syncthreads();
            }
            if (lane < D) demb[(long long)b * fd + i * D + lane] = tot;
        }
    }
    // lane partials -> one row per workgroup (fixed shuffle order)
    float* prow = partials + (long long)blockIdx.x * (A * D + 2 * A + D + 1);
#pragma unroll
    for (int i = 0; i < A * D; ++i) {
        const float t = rs_wave_sum(accW[i]);
        if (lane == 0) prow[i] = t;
    }
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const float tb = rs_wave_sum(accb[i]), th = rs_wave_sum(acch[i]);
        if (lane == 0) { prow[A * D + i] = tb; prow[A * D + A + i] = th; }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float t = rs_wave_sum(accpw[i]);
        if (lane == 0) prow[A * D + 2 * A + i] = t;
    }
    const float tpb = rs_wave_sum(accpb);
    if (lane == 0) prow[A * D + 2 * A + D] = tpb;
}


// ---- FFM with true field-aware tables (model/FFM/FFM.py:43-62) -------------------------------------
// Stacked table T [R, F*d]: row (row_offset[i] + x_i) holds the F vectors of feature value x_i, one per
// partner field.  out[b] = sum_{i<j} < T[r_i][j], T[r_j][i] >.  One wave per sample, lanes over pairs: the
// reads T[r_i][j], j = i+1.. of neighbouring lanes are contiguous.
__global__ __launch_bounds__(256) void ffm_fwd_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ row_offset,
                                                      const int64_t* __restrict__ vocab, const float* __restrict__ T,
                                                      int B, int F, int d, float* __restrict__ out,
                                                      int* __restrict__ err_flag) {
    extern __shared__ float sm[];
    long long* rows = reinterpret_cast<long long*>(sm) + (threadIdx.x >> 6) * F;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool valid = b < B;
    if (valid) {
        for (int f = lane; f < F; f += 64) {
            long long x = idx[(long long)f * B + b];
            if (x < 0 || x >= vocab[f]) { x = 0; if (err_flag) atomicOr(err_flag, 1); }
            rows[f] = row_offset[f] + x;
        }
    }
    __syncthreads();
    if (!valid) return;
    const int P = F * (F - 1) / 2;
    const long long fd = (long long)F * d;
    float acc = 0.f;
    for (int p = lane; p < P; p += 64) {
        int i, j;
        pair_decode(p, F, &i, &j);
        const float* a = T + rows[i] * fd + (long long)j * d;
        const float* c = T + rows[j] * fd + (long long)i * d;
        float t = 0.f;
        for (int k = 0; k < d; ++k) t = fmaf(a[k], c[k], t);
        acc += t;
    }
    acc = rs_wave_sum(acc);
    if (lane == 0) out[b] = acc;
}

// Dense gradient of the stacked FFM table.  One workgroup per distinct row (a segment of the sorted batch,
// see sparse.hip): it walks the row's samples in ascending order and sums
//   dT[r_i][j*d+k] += dout[b] * T[r_j(b)][i*d+k]   (j != i)
// in registers (fixed order, no atomics), then writes the row once.  Rows outside the batch and the padding
// row keep the caller's zeros.
__global__ __launch_bounds__(256) void ffm_bwd_kernel(PlanView pv, const int64_t* __restrict__ idx,
                                                      const int64_t* __restrict__ row_offset, const int64_t* __restrict__ vocab,
                                                      const float* __restrict__ T, const float* __restrict__ dout, int B,
                                                      int F, int cap, int d, float* __restrict__ dT) {
    const int f = blockIdx.y, s = blockIdx.x;
    if (s >= pv.counts[f * 4 + 2]) return;
    const int* sfc = pv.seg_first_chunk + (long long)f * (cap + 1);
    const int* cst = pv.chunk_start + (long long)f * (cap + 1);
    const int beg = cst[sfc[s]], end = cst[sfc[s + 1]];
    const long long fd = (long long)F * d;
    const long long row = pv.seg_row[(long long)f * cap + s];
    for (int c = threadIdx.x; c < F * d; c += 256) {
        const int j = c / d, k = c - j * d;
        float acc = 0.f;
        if (j != f) {
            const long long V = vocab[j], ro = row_offset[j];
            for (int q = beg; q < end; ++q) {
                const int b = pv.sorted_src[(long long)f * cap + q] / F;
                long long x = idx[(long long)j * B + b];
                if (x < 0 || x >= V) x = 0;
                acc = fmaf(dout[b], T[(ro + x) * fd + (long long)f * d + k], acc);
            }
        }
        dT[row * fd + c] = acc;
    }
}

constexpr int LDS_LIMIT = 64 * 1024;
constexpr int PARTIAL_ROWS_MAX = 512;

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


// Rows of the per-workgroup partial buffers of the *_bwd entry points below for a batch of B samples.
int rs_interact_partial_rows(int B) { return B < 1 ? 1 : (B < PARTIAL_ROWS_MAX ? B : PARTIAL_ROWS_MAX); }

// PNN inner products: emb [B,F,d] -> out [B, F(F-1)/2], pairs (i<j) row-major (model/PNN/PNN.py:65-77).
int rs_pnn_inner_fwd(const float* emb, int B, int F, int d, float* out, hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 1 && d > 0, "rs_pnn_inner_fwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(emb && out, "rs_pnn_inner_fwd: null argument");
    const size_t lds = (size_t)4 * F * d * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_pnn_inner_fwd: F*d=%d does not fit the LDS staging", F * d);
    hipLaunchKernelGGL(pnn_inner_fwd_kernel, dim3(rs_cdiv(B, 4)), dim3(256), lds, stream, emb, B, F, d, out);
    RS_CHECK_LAUNCH("rs_pnn_inner_fwd");
    return RS_OK;
}

int rs_pnn_inner_bwd(const float* emb, const float* dout, int B, int F, int d, float* demb, hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 1 && d > 0, "rs_pnn_inner_bwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(emb && dout && demb, "rs_pnn_inner_bwd: null argument");
    const size_t lds = (size_t)4 * (F * d + F * (F - 1) / 2) * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_pnn_inner_bwd: F=%d d=%d does not fit the LDS staging", F, d);
    hipLaunchKernelGGL(pnn_inner_bwd_kernel, dim3(rs_cdiv(B, 4)), dim3(256), lds, stream, emb, dout, B, F, d, demb);
    RS_CHECK_LAUNCH("rs_pnn_inner_bwd");
    return RS_OK;
}

// PNN outer products: emb [B,F,d] -> out [B, P*d*d] (model/PNN/PNN.py:80-103).
int rs_pnn_outer_fwd(const float* emb, int B, int F, int d, float* out, hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 1 && F < 32768 && d > 0, "rs_pnn_outer_fwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(emb && out, "rs_pnn_outer_fwd: null argument");
    const int P = F * (F - 1) / 2;
    RS_CHECK_ARG((long long)P * d * d < (1ll << 31), "rs_pnn_outer_fwd: output row too long");
    const size_t lds = (size_t)(F * d + P) * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_pnn_outer_fwd: F=%d d=%d does not fit the LDS staging", F, d);
    if (d % 4 == 0) hipLaunchKernelGGL(pnn_outer_fwd_kernel<4>, dim3(B), dim3(256), lds, stream, emb, B, F, d, out);
    else hipLaunchKernelGGL(pnn_outer_fwd_kernel<1>, dim3(B), dim3(256), lds, stream, emb, B, F, d, out);
    RS_CHECK_LAUNCH("rs_pnn_outer_fwd");
    return RS_OK;
}

int rs_pnn_outer_bwd(const float* emb, const float* dout, int B, int F, int d, float* demb, hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 1 && d > 0, "rs_pnn_outer_bwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(emb && dout && demb, "rs_pnn_outer_bwd: null argument");
    const size_t lds = (size_t)F * d * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_pnn_outer_bwd: F*d=%d does not fit the LDS staging", F * d);
    hipLaunchKernelGGL(pnn_outer_bwd_kernel, dim3(B), dim3(256), lds, stream, emb, dout, B, F, d, demb);
    RS_CHECK_LAUNCH("rs_pnn_outer_bwd");
    return RS_OK;
}

// DCN cross network, L layers: x0 [B,D], w [L,D], bias [L,D] -> out [B,D]; S [B,L] (x_l . w_l) is saved
// for rs_dcn_cross_bwd (model/DCN/DCN.py:48-52).
int rs_dcn_cross_fwd(const float* x0, const float* w, const float* bias, int B, int D, int L, float* out, float* S,
                     hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && D > 0 && L > 0, "rs_dcn_cross_fwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(x0 && w && bias && out, "rs_dcn_cross_fwd: null argument");
    const size_t lds = (size_t)4 * D * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_dcn_cross_fwd: D=%d does not fit the LDS staging", D);
    hipLaunchKernelGGL(dcn_cross_fwd_kernel, dim3(rs_cdiv(B, 4)), dim3(256), lds, stream, x0, w, bias, B, D, L, out, S);
    RS_CHECK_LAUNCH("rs_dcn_cross_fwd");
    return RS_OK;
}

// dx0 [B,D]; partials [rows, 2*L*D] with rows = rs_interact_partial_rows(B): per-row [dW | dB] partial sums,
// to be summed over rows by the caller.
int rs_dcn_cross_bwd(const float* x0, const float* S, const float* dout, const float* w, const float* bias, int B, int D,
                     int L, float* dx0, float* partials, int rows, hipStream_t stream) {
    RS_CHECK_ARG(B > 0 && D > 0 && L > 0, "rs_dcn_cross_bwd: bad sizes");
    RS_CHECK_ARG(x0 && S && dout && w && bias && dx0 && partials, "rs_dcn_cross_bwd: null argument");
    RS_CHECK_ARG(rows == rs_interact_partial_rows(B), "rs_dcn_cross_bwd: partials must have %d rows",
                 rs_interact_partial_rows(B));
    const size_t lds = (size_t)(3 * L + 2) * D * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_dcn_cross_bwd: L=%d D=%d does not fit the LDS staging", L, D);
    hipLaunchKernelGGL(dcn_cross_bwd_kernel, dim3(rows), dim3(64), lds, stream, x0, S, dout, w, bias, B, D, L, dx0, partials);
    RS_CHECK_LAUNCH("rs_dcn_cross_bwd");
    return RS_OK;
}

// NFM bi-interaction, reference form: E [B,D], V [D,K] -> out [B,K]; s_out [B,K] = E V is saved for the
// backward (model/NFM/NFM.py:42-44).
int rs_nfm_bi_fwd(const float* E, const float* V, int B, int D, int K, float* out, float* s_out, hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && D > 0 && K > 0 && K <= NFM_KMAX, "rs_nfm_bi_fwd: bad sizes (hidden_dim <= %d)", NFM_KMAX);
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(E && V && out, "rs_nfm_bi_fwd: null argument");
    hipLaunchKernelGGL(nfm_bi_fwd_kernel, dim3(rs_cdiv(B, 4)), dim3(256), 0, stream, E, V, B, D, K, out, s_out);
    RS_CHECK_LAUNCH("rs_nfm_bi_fwd");
    return RS_OK;
}

// dE [B,D]; partials [rows, D*K] per-row partial dV.
int rs_nfm_bi_bwd(const float* E, const float* V, const float* s, const float* dout, int B, int D, int K, float* dE,
                  float* partials, int rows, hipStream_t stream) {
    RS_CHECK_ARG(B > 0 && D > 0 && K > 0 && K <= NFM_KMAX, "rs_nfm_bi_bwd: bad sizes (hidden_dim <= %d)", NFM_KMAX);
    RS_CHECK_ARG(E && V && s && dout && dE && partials, "rs_nfm_bi_bwd: null argument");
    RS_CHECK_ARG(rows == rs_interact_partial_rows(B), "rs_nfm_bi_bwd: partials must have %d rows",
                 rs_interact_partial_rows(B));
    const size_t lds = (size_t)D * K * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_nfm_bi_bwd: D*K=%d does not fit the LDS staging", D * K);
    hipLaunchKernelGGL(nfm_bi_bwd_kernel, dim3(rows), dim3(64), lds, stream, E, V, s, dout, B, D, K, dE, partials);
    RS_CHECK_LAUNCH("rs_nfm_bi_bwd");
    return RS_OK;
}

// AFM attention pooling: emb [B,F,d]; attention W [A,d], bA [A], h [A]; projection pw [d], pb [1] -> out [B]
// (model/AFM/AFM.py:19-24,36-42,47-56).  d in {4,8,16,32}.
int rs_afm_fwd(const float* emb, int B, int F, int d, int A, const float* W, const float* bA, const float* h,
               const float* pw, const float* pb, float* out, hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 1 && A > 0, "rs_afm_fwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(emb && W && bA && h && pw && pb && out, "rs_afm_fwd: null argument");
    const int P = F * (F - 1) / 2;
    const size_t lds = (size_t)(A * d + 2 * A + d + F * d + P) * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_afm_fwd: F=%d d=%d A=%d does not fit the LDS staging", F, d, A);
    const int grid = B < 2048 ? B : 2048;
#define RS_AFM_F(DD) hipLaunchKernelGGL(afm_fwd_kernel<DD>, dim3(grid), dim3(64), lds, stream, emb, B, F, A, W, bA, h, pw, pb, out)
    switch (d) {
        case 4: RS_AFM_F(4); break;
        case 8: RS_AFM_F(8); break;
        case 16: RS_AFM_F(16); break;
        case 32: RS_AFM_F(32); break;
        default: RS_CHECK_ARG(false, "rs_afm_fwd: embedding_dim %d not in {4,8,16,32}", d);
    }
#undef RS_AFM_F
    RS_CHECK_LAUNCH("rs_afm_fwd");
    return RS_OK;
}

// demb [B,F,d]; partials [rows, A*d + 2A + d + 1] per-row partial [dW | dbA | dh | dpw | dpb].
// d in {4,8,16,32}, A in {4,8,16}, A*d <= 256.
int rs_afm_bwd(const float* emb, const float* dout, int B, int F, int d, int A, const float* W, const float* bA,
               const float* h, const float* pw, float* demb, float* partials, int rows, hipStream_t stream) {
    RS_CHECK_ARG(B > 0 && F > 1, "rs_afm_bwd: bad sizes");
    RS_CHECK_ARG(emb && dout && W && bA && h && pw && demb && partials, "rs_afm_bwd: null argument");
    RS_CHECK_ARG(rows == rs_interact_partial_rows(B), "rs_afm_bwd: partials must have %d rows", rs_interact_partial_rows(B));
    const int P = F * (F - 1) / 2;
    const size_t lds = (size_t)(A * d + 2 * A + d + F * d + 2 * P + 64 * (d + 1)) * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_afm_bwd: F=%d d=%d A=%d does not fit the LDS staging", F, d, A);
#define RS_AFM_B(DD, AA)                                                                                              \
    hipLaunchKernelGGL((afm_bwd_kernel<DD, AA>), dim3(rows), dim3(64), lds, stream, emb, dout, B, F, W, bA, h, pw, demb, \
                       partials)
    const int key = d * 100 + A;
    switch (key) {
        case 404: RS_AFM_B(4, 4); break;
        case 408: RS_AFM_B(4, 8); break;
        case 416: RS_AFM_B(4, 16); break;
        case 804: RS_AFM_B(8, 4); break;
        case 808: RS_AFM_B(8, 8); break;
        case 816: RS_AFM_B(8, 16); break;
        case 1604: RS_AFM_B(16, 4); break;
        case 1608: RS_AFM_B(16, 8); break;
        case 1616: RS_AFM_B(16, 16); break;
        case 3204: RS_AFM_B(32, 4); break;
        case 3208: RS_AFM_B(32, 8); break;
        default: RS_CHECK_ARG(false, "rs_afm_bwd: (embedding_dim %d, attention_dim %d) not supported", d, A);
    }
#undef RS_AFM_B
    RS_CHECK_LAUNCH("rs_afm_bwd");
    return RS_OK;
}


// FFM second-order term with field-aware tables: idx [F,B] int64 columnar, T [sum V_f, F*d] -> out [B]
// (model/FFM/FFM.py:43-62 with independent tables per (feature, partner field)).
int rs_ffm_fwd(const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, const float* T, int B, int F, int d,
               float* out, int* err_flag, hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 1 && d > 0, "rs_ffm_fwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(idx && row_offset && vocab && T && out, "rs_ffm_fwd: null argument");
    const size_t lds = (size_t)4 * F * sizeof(long long);
    hipLaunchKernelGGL(ffm_fwd_kernel, dim3(rs_cdiv(B, 4)), dim3(256), lds, stream, idx, row_offset, vocab, T, B, F, d, out,
                       err_flag);
    RS_CHECK_LAUNCH("rs_ffm_fwd");
    return RS_OK;
}

// Dense table gradient dT [sum V_f, F*d] (caller-zeroed) from dout [B].  plan: rs_sparse_bwd_sort of the same
// columnar idx with capacity cap (entry (f, b) carrying gradient-row index b*F+f).
int rs_ffm_bwd(const int* plan, const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, const float* T,
               const float* dout, int B, int F, int cap, int d, float* dT, hipStream_t stream) {
    RS_CHECK_ARG(B > 0 && F > 1 && d > 0 && cap >= B, "rs_ffm_bwd: bad sizes");
    RS_CHECK_ARG(plan && idx && row_offset && vocab && T && dout && dT, "rs_ffm_bwd: null argument");
    const PlanView pv = carve_plan(const_cast<int*>(plan), F, cap);
    hipLaunchKernelGGL(ffm_bwd_kernel, dim3(B, F), dim3(256), 0, stream, pv, idx, row_offset, vocab, T, dout, B, F, cap, d, dT);
    RS_CHECK_LAUNCH("rs_ffm_bwd");
    return RS_OK;
}

}  // extern "C"
