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

// Matrix-core form of the two kernels above for F <= 64 fields and d in {4, 8, 16, 32}: the inner products of a
// sample are the upper triangle of its Gram matrix E E^T (E = [F, d]), the gradient is Gs E with Gs the symmetric
// zero-diagonal matrix holding dP.  One wave per sample, v_mfma_f32_16x16x4_f32 on 16-field blocks.
// Operand convention of the instruction: lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16]; it receives
// C[4 * (l / 16) + v][l % 16], v = 0..3.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Forward: lane (r, g) loads the float4 E[16I + r][4(g + 4q) ..] of every field block I (one coalesced 1 KB read per
// block); component s of that float4 serves as the lane's k-slot of MFMA step (q, s) for the A and the B operand
// alike (the summation index may be visited in any order as long as both operands agree).
// ld: row stride of out.  CAT: a row of out is [the sample's F*D embedding values | its P products | zeros up to ld]
// -- the input of PNN's first MLP layer (PNN.py:58 with pnn_type inner), its width rounded up to four floats -- so
// that neither the concatenation nor the padding needs a launch of its own.
template <int NB, int D, bool CAT>
__global__ __launch_bounds__(256) void pnn_inner_fwd_mfma_kernel(const float* __restrict__ emb, int B, int F,
                                                                 float* __restrict__ out, long long ld) {
    constexpr int NV = D >= 16 ? D / 16 : 1;
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int nwaves = gridDim.x * 4;
    const int fd = F * D, P = F * (F - 1) / 2;
    for (int b = blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += nwaves) {
        const float* e = emb + (long long)b * fd;
        float* orow = out + (long long)b * ld;
        float v[NB][NV][4];
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int row = 16 * I + r, c = g + 4 * q;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < F && c < D / 4) {
                    t = *reinterpret_cast<const float4*>(e + row * D + 4 * c);
                    if (CAT) *reinterpret_cast<float4*>(orow + row * D + 4 * c) = t;
                }
                v[I][q][0] = t.x; v[I][q][1] = t.y; v[I][q][2] = t.z; v[I][q][3] = t.w;
            }
        if (CAT && lane < (int)(ld - fd - P)) orow[fd + P + lane] = 0.f;
        float* o = orow + (CAT ? fd : 0);
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = I; J < NB; ++J) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < NV; ++q)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[I][q][s4], v[J][q][s4], acc, 0, 0, 0);
                const int j = 16 * J + r;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int i = 16 * I + 4 * g + t;
                    if (i < j && j < F) o[pair_start(i, F) + j - i - 1] = acc[t];
                }
            }
    }
}

// Backward: demb = Gs E.  The sample's dP row is copied to LDS with coalesced reads; the A operand of step s, row
// block I -- Gs[16I + r][4s + g] -- is gathered from there through pair slots that do not depend on the sample
// (computed once per wave); B operand: E[4s + g][16c + r], straight from HBM (256 contiguous bytes per request).
// ld: row stride of dout.  CAT: a row of dout is the gradient of the concatenated row the forward writes; its first
// F*D values are added to the result (the embeddings' own path into the MLP).
template <int NB, int D, bool CAT>
__global__ __launch_bounds__(256) void pnn_inner_bwd_mfma_kernel(const float* __restrict__ emb, const float* __restrict__ dout,
                                                                 int B, int F, float* __restrict__ demb, long long ld) {
    extern __shared__ float sm[];
    constexpr int KS = NB * 4;
    constexpr int NC = D >= 16 ? D / 16 : 1;
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int nwaves = gridDim.x * 4;
    const int fd = F * D, P = F * (F - 1) / 2;
    float* gl = sm + (threadIdx.x >> 6) * (P + 1);         // this wave's dP row; slot P holds 0 (diagonal / padding)
    int slot[NB][KS];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4) {
            const int i = 16 * I + r, k = 4 * s4 + g;
            slot[I][s4] = (i < F && k < F && i != k) ? pair_index_any(i, k, F) : P;
        }
    if (lane == 0) gl[P] = 0.f;
    // A wave walks its samples one after the other, and a sample is memory latency followed by LDS gathers and matrix
    // instructions: the next sample's dP row and E operands are requested (into registers) before the current one is
    // worked on.
    constexpr int NPR = (16 * NB * (16 * NB - 1) / 2 + 63) / 64;       // dP values per lane, at most
    float pg[NPR], pbv[KS][NC];
    auto fetch = [&](int bb) {
        const float* e = emb + (long long)bb * fd;
        const float* gs = dout + (long long)bb * ld + (CAT ? fd : 0);
#pragma unroll
        for (int i = 0; i < NPR; ++i) pg[i] = (lane + 64 * i < P) ? gs[lane + 64 * i] : 0.f;
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int k = 4 * s4 + g, col = 16 * c + r;
                pbv[s4][c] = (k < F && col < D) ? e[k * D + col] : 0.f;
            }
    };
    const int b_first = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b_first < B) fetch(b_first);
    for (int b = b_first; b < B; b += nwaves) {
        const float* drow = dout + (long long)b * ld;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NPR; ++i)
            if (lane + 64 * i < P) gl[lane + 64 * i] = pg[i];
        float bv[KS][NC];
#pragma unroll
        for (int s4 = 0; s4 < KS; ++s4)
#pragma unroll
            for (int c = 0; c < NC; ++c) bv[s4][c] = pbv[s4][c];
        if (b + nwaves < B) fetch(b + nwaves);
        __builtin_amdgcn_wave_barrier();
        float* o = demb + (long long)b * fd;
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            f32x4 acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < KS; ++s4) {
                if (4 * s4 < F) {
                    const float a = gl[slot[I][s4]];
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[s4][c], acc[c], 0, 0, 0);
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = 16 * I + 4 * g + t, col = 16 * c + r;
                    if (row < F && col < D) o[row * D + col] = CAT ? acc[c][t] + drow[row * D + col] : acc[c][t];
                }
        }
    }
}

// ---- PNN outer product (model/PNN/PNN.py:80-103) --------------------------------------------------
// out[b, p, a, c] = e_i[a] * e_j[c] for pair p = (i, j).  One workgroup per sample, the (i, j) table of
// all pairs in LDS; the kernel is bound by the HBM write of P*d*d floats per sample.
// The kernels take a range of pairs [p_lo, p_hi) (a range of first fields: pairs are numbered first field major) and the
// row stride of out / dout, whose column 0 is pair p_lo: the first MLP layer over the products can then be evaluated a
// block of pairs at a time without the whole (B, P*d*d) matrix ever existing.
template <int VEC>
__global__ __launch_bounds__(256) void pnn_outer_fwd_kernel(const float* __restrict__ emb, int B, int F, int d,
                                                            float* __restrict__ out, int p_lo, int p_hi, long long ld) {
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
    float* o = out + (long long)b * ld;
    const int total = (p_hi - p_lo) * dd / VEC;
    for (int q = threadIdx.x; q < total; q += 256) {
        const int x = q * VEC;
        const int pl = x / dd, r = x - pl * dd;
        const int a = r / d, c = r - a * d;
        const int ij = tab[p_lo + pl], i = ij & 0xffff, j = ij >> 16;
        const float ea = e[i * d + a];
        if constexpr (VEC == 4) {
            const float4 ej = *reinterpret_cast<const float4*>(e + j * d + c);
            *reinterpret_cast<float4*>(o + x) = make_float4(ea * ej.x, ea * ej.y, ea * ej.z, ea * ej.w);
        } else {
            o[x] = ea * e[j * d + c];
        }
    }
}

// d = 16: index arithmetic by shifts; a wave writes one whole 16x16 block (1 KB contiguous) per step.
__global__ __launch_bounds__(256) void pnn_outer_fwd16_kernel(const float* __restrict__ emb, int B, int F,
                                                              float* __restrict__ out, int p_lo, int p_hi, long long ld) {
    extern __shared__ float sm[];
    const int fd = F * 16, P = F * (F - 1) / 2;
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
    float4* o = reinterpret_cast<float4*>(out + (long long)b * ld);
    const int total = (p_hi - p_lo) * 64;
#pragma unroll 4
    for (int q = threadIdx.x; q < total; q += 256) {
        const int p = p_lo + (q >> 6), a = (q >> 2) & 15, c4 = q & 3;
        const int ij = tab[p], i = ij & 0xffff, j = ij >> 16;
        const float ea = e[i * 16 + a];
        const float4 ej = *reinterpret_cast<const float4*>(e + j * 16 + 4 * c4);
        o[q] = make_float4(ea * ej.x, ea * ej.y, ea * ej.z, ea * ej.w);
    }
}

// demb[b,f,k] = sum_{j>f} sum_c dO[p(f,j), k, c] e_j[c] + sum_{i<f} sum_a dO[p(i,f), a, k] e_i[a]
__global__ __launch_bounds__(256) void pnn_outer_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ dout,
                                                            int B, int F, int d, float* __restrict__ demb, int i_lo, int i_hi,
                                                            long long ld, int accumulate) {
    extern __shared__ float sm[];
    const int fd = F * d, dd = d * d;
    const int p_lo = pair_start(i_lo, F);
    float* e = sm;
    const int b = blockIdx.x;
    const float* src = emb + (long long)b * fd;
    for (int i = threadIdx.x; i < fd; i += 256) e[i] = src[i];
    __syncthreads();
    const float* g = dout + (long long)b * ld - (long long)p_lo * dd;
    for (int idx = threadIdx.x; idx < fd; idx += 256) {
        const int f = idx / d, k = idx - f * d;
        float acc = 0.f;
        if (f >= i_lo && f < i_hi)
            for (int j = f + 1; j < F; ++j) {
                const float* row = g + (long long)pair_index(f, j, F) * dd + k * d;
                for (int c = 0; c < d; ++c) acc = fmaf(row[c], e[j * d + c], acc);
            }
        for (int i = i_lo; i < min(f, i_hi); ++i) {
            const float* col = g + (long long)pair_index(i, f, F) * dd + k;
            for (int a = 0; a < d; ++a) acc = fmaf(col[a * d], e[i * d + a], acc);
        }
        float* dst = demb + (long long)b * fd + idx;
        *dst = accumulate ? *dst + acc : acc;
    }
}

// The same gradient for d = 16 with every dO element read once through full 64-byte requests.  A 16x16 block of dO
// is held by 16 lanes as 4x4 sub-tiles (lane = (group, ta, tc): rows 4ta.., columns 4tc..), four blocks (i, j..j+3)
// per wave and step.  Row sums (for e_i) accumulate in registers over j and are reduced across lanes once per i;
// column sums (for e_j) are reduced over ta with two shuffle steps and added to the wave's own LDS accumulator.
// The four waves of the workgroup (rows i = wave, wave + 4, ...) are combined in fixed order at the end.
__global__ __launch_bounds__(256) void pnn_outer_bwd16_kernel(const float* __restrict__ emb, const float* __restrict__ dout,
                                                              int B, int F, float* __restrict__ demb, int i_lo, int i_hi,
                                                              long long ld, int accumulate) {
    extern __shared__ float sm[];
    const int fd = F * 16;
    float* e = sm;                    // [F*16]
    float* rowres = e + fd;           // [F*16]
    float* colacc = rowres + fd;      // [4][F*16]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float* src = emb + (long long)b * fd;
    for (int i = threadIdx.x; i < fd; i += 256) { e[i] = src[i]; rowres[i] = 0.f; }
    for (int i = threadIdx.x; i < 4 * fd; i += 256) colacc[i] = 0.f;
    __syncthreads();
    const int tc = lane & 3, ta = (lane >> 2) & 3, gq = lane >> 4;
    const float* gbase = dout + (long long)b * ld - (long long)pair_start(i_lo, F) * 256;
    float* myc = colacc + wv * fd;
    for (int i = i_lo + wv; i < min(i_hi, F - 1); i += 4) {
        float ei[4], racc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { ei[r] = e[i * 16 + 4 * ta + r]; racc[r] = 0.f; }
        const int p0 = pair_start(i, F) - i - 1;           // block of pair (i, j) = p0 + j
        for (int j0 = i + 1; j0 < F; j0 += 4) {
            const int j = j0 + gq;
            const bool on = j < F;
            float4 v[4];
            float4 ej = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on) {
                const float* blk = gbase + (long long)(p0 + j) * 256 + 4 * tc;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = *reinterpret_cast<const float4*>(blk + (4 * ta + r) * 16);
                ej = *reinterpret_cast<const float4*>(e + j * 16 + 4 * tc);
            }
            float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                racc[r] = fmaf(v[r].x, ej.x, racc[r]); racc[r] = fmaf(v[r].y, ej.y, racc[r]);
                racc[r] = fmaf(v[r].z, ej.z, racc[r]); racc[r] = fmaf(v[r].w, ej.w, racc[r]);
                cp.x = fmaf(v[r].x, ei[r], cp.x); cp.y = fmaf(v[r].y, ei[r], cp.y);
                cp.z = fmaf(v[r].z, ei[r], cp.z); cp.w = fmaf(v[r].w, ei[r], cp.w);
            }
#pragma unroll
            for (int o = 4; o <= 8; o <<= 1) {
                cp.x += __shfl_xor(cp.x, o, 64); cp.y += __shfl_xor(cp.y, o, 64);
                cp.z += __shfl_xor(cp.z, o, 64); cp.w += __shfl_xor(cp.w, o, 64);
            }
            if (on && ta == 0) {
                float4* dst = reinterpret_cast<float4*>(myc + j * 16 + 4 * tc);
                float4 t = *dst;
                t.x += cp.x; t.y += cp.y; t.z += cp.z; t.w += cp.w;
                *dst = t;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = racc[r];
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
            if (tc == 0 && gq == 0) rowres[i * 16 + 4 * ta + r] = t;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < fd; i += 256) {
        const float t = rowres[i] + (((colacc[i] + colacc[fd + i]) + colacc[2 * fd + i]) + colacc[3 * fd + i]);
        float* dst = demb + (long long)b * fd + i;
        *dst = accumulate ? *dst + t : t;
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

// Backward of the cross network.  A wave walks samples; element n lives in lane n % 64 throughout.  Parameter
// gradients accumulate in lane-private LDS words of the wave's own region, x_l is recomputed from x0 and S where it
// is needed, and the waves of a workgroup are summed in fixed order into one partial row:
// partials[g] = [dW (L*D) | dB (L*D)], summed over g by the caller in fixed order.
__global__ __launch_bounds__(256) void dcn_cross_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ S,
                                                            const float* __restrict__ dout, const float* __restrict__ w,
                                                            const float* __restrict__ bias, int B, int D, int L,
                                                            float* __restrict__ dx0, float* __restrict__ partials,
                                                            const float* __restrict__ dxadd) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int region = (2 * L + 2) * D;
    float* accW = sm + wv * region;   // [L][D]
    float* accB = accW + L * D;       // [L][D]
    float* dx = accB + L * D;         // [D]
    float* d0 = dx + D;               // [D]
    for (int i = lane; i < 2 * L * D; i += 64) accW[i] = 0.f;
    for (int b = blockIdx.x * nw + wv; b < B; b += gridDim.x * nw) {
        const float* xs = x0 + (long long)b * D;
        const float* sb = S + (long long)b * L;
        float part = 0.f;
        for (int n = lane; n < D; n += 64) {
            const float g = dout[(long long)b * D + n];
            dx[n] = g;
            d0[n] = 0.f;
            part = fmaf(g, xs[n], part);
        }
        for (int l = L - 1; l >= 0; --l) {
            const float ds = rs_wave_sum(part);
            const float s = sb[l];
            part = 0.f;
            for (int n = lane; n < D; n += 64) {
                const float x0n = xs[n];
                float xl = x0n;
                for (int t = 0; t < l; ++t) xl = x0n * sb[t] + bias[t * D + n] + xl;
                const float g = dx[n];
                accB[l * D + n] += g;
                accW[l * D + n] = fmaf(ds, xl, accW[l * D + n]);
                d0[n] = fmaf(g, s, d0[n]);
                const float gn = fmaf(ds, w[l * D + n], g);
                dx[n] = gn;
                part = fmaf(gn, x0n, part);
            }
        }
        for (int n = lane; n < D; n += 64)
            dx0[(long long)b * D + n] = d0[n] + dx[n] + (dxadd ? dxadd[(long long)b * D + n] : 0.f);
    }
    __syncthreads();
    float* prow = partials + (long long)blockIdx.x * 2 * L * D;
    for (int i = threadIdx.x; i < 2 * L * D; i += blockDim.x) {
        float t = 0.f;
        for (int v = 0; v < nw; ++v) t += sm[v * region + i];
        prow[i] = t;
    }
}

// Register forms of the two cross-network kernels for D <= 64*T and L <= 4 layers: a lane keeps its T elements of
// x0, x_l (forward) or of dx, d0 and of the 2*L parameter-gradient accumulators (backward) in registers; w and b sit
// in LDS once per workgroup.  Same per-element arithmetic and the same summation orders as the LDS kernels above.
template <int T, int L>
__global__ __launch_bounds__(256) void dcn_cross_fwd_reg_kernel(const float* __restrict__ x0, const float* __restrict__ w,
                                                                const float* __restrict__ bias, int B, int D,
                                                                float* __restrict__ out, float* __restrict__ S) {
    extern __shared__ float sm[];
    float* sw = sm;
    float* sbias = sm + L * D;
    for (int i = threadIdx.x; i < L * D; i += 256) { sw[i] = w[i]; sbias[i] = bias[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int b = blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += gridDim.x * 4) {
        const float* xs = x0 + (long long)b * D;
        float x0r[T], x[T];
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int n = lane + 64 * i;
            x0r[i] = n < D ? xs[n] : 0.f;
            x[i] = x0r[i];
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const int n = lane + 64 * i;
                if (n < D) part = fmaf(x[i], sw[l * D + n], part);
            }
            const float s = rs_wave_sum(part);
            if (lane == 0 && S) S[(long long)b * L + l] = s;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const int n = lane + 64 * i;
                if (n < D) x[i] = x0r[i] * s + sbias[l * D + n] + x[i];
            }
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int n = lane + 64 * i;
            if (n < D) out[(long long)b * D + n] = x[i];
        }
    }
}

template <int T, int L>
__global__ __launch_bounds__(256) void dcn_cross_bwd_reg_kernel(const float* __restrict__ x0, const float* __restrict__ S,
                                                                const float* __restrict__ dout, const float* __restrict__ w,
                                                                const float* __restrict__ bias, int B, int D,
                                                                float* __restrict__ dx0, float* __restrict__ partials,
                                                                const float* __restrict__ dxadd) {
    extern __shared__ float sm[];        // first w | bias, at the end the four waves' accumulators [4][2*L*D]
    float* sw = sm;
    float* sbias = sm + L * D;
    for (int i = threadIdx.x; i < L * D; i += 256) { sw[i] = w[i]; sbias[i] = bias[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float accW[L][T], accB[L][T];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int i = 0; i < T; ++i) accW[l][i] = accB[l][i] = 0.f;
    // The operands of the wave's next sample are requested before the current one is worked on: a sample is two
    // dependent phases (memory latency, then L wave reductions), and a wave walks its samples one after the other.
    const int bstep = gridDim.x * 4;
    float nsv[L], nx0[T], ndx[T];
    auto fetch = [&](int bb) {
#pragma unroll
        for (int l = 0; l < L; ++l) nsv[l] = S[(long long)bb * L + l];
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int n = lane + 64 * i;
            nx0[i] = n < D ? x0[(long long)bb * D + n] : 0.f;
            ndx[i] = n < D ? dout[(long long)bb * D + n] : 0.f;
        }
    };
    const int b_first = blockIdx.x * 4 + wv;
    if (b_first < B) fetch(b_first);
    for (int b = b_first; b < B; b += bstep) {
        float sv[L];
#pragma unroll
        for (int l = 0; l < L; ++l) sv[l] = nsv[l];
        float x0r[T], dx[T], d0[T];
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            x0r[i] = nx0[i];
            dx[i] = ndx[i];
            d0[i] = 0.f;
            part = fmaf(dx[i], x0r[i], part);        // lanes past D hold zeros
        }
        if (b + bstep < B) fetch(b + bstep);
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            const float ds = rs_wave_sum(part);
            const float s = sv[l];
            part = 0.f;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const int n = lane + 64 * i;
                if (n < D) {
                    float xl = x0r[i];
#pragma unroll
                    for (int t = 0; t < l; ++t) xl = x0r[i] * sv[t] + sbias[t * D + n] + xl;
                    const float g = dx[i];
                    accB[l][i] += g;
                    accW[l][i] = fmaf(ds, xl, accW[l][i]);
                    d0[i] = fmaf(g, s, d0[i]);
                    const float gn = fmaf(ds, sw[l * D + n], g);
                    dx[i] = gn;
                    part = fmaf(gn, x0r[i], part);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int n = lane + 64 * i;
            if (n < D) dx0[(long long)b * D + n] = d0[i] + dx[i] + (dxadd ? dxadd[(long long)b * D + n] : 0.f);
        }
    }
    __syncthreads();                     // every wave is done with w | bias
    float* mine = sm + wv * 2 * L * D;
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int n = lane + 64 * i;
            if (n < D) { mine[l * D + n] = accW[l][i]; mine[L * D + l * D + n] = accB[l][i]; }
        }
    __syncthreads();
    float* prow = partials + (long long)blockIdx.x * 2 * L * D;
    for (int i = threadIdx.x; i < 2 * L * D; i += 256)
        prow[i] = ((sm[i] + sm[2 * L * D + i]) + sm[4 * L * D + i]) + sm[6 * L * D + i];
}

// ---- NFM bi-interaction in the reference's form (model/NFM/NFM.py:42-44) ---------------------------
// out[b,c] = 0.5 * ((sum_n e_n v_nc)^2 - sum_n e_n^2 v_nc^2), e = flattened (B, D) embeddings, v (D, K).
constexpr int NFM_KMAX = 16;
template <int KT>
__global__ __launch_bounds__(256) void nfm_bi_fwd_kernel(const float* __restrict__ E, const float* __restrict__ V, int B,
                                                         int D, int K, float* __restrict__ out, float* __restrict__ s_out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float s[KT], q[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) s[c] = q[c] = 0.f;
    for (int n = lane; n < D; n += 64) {
        const float e = E[(long long)b * D + n];
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            if (c < K) {
                const float v = V[n * K + c];
                s[c] = fmaf(e, v, s[c]);
                q[c] = fmaf(e * e, v * v, q[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < KT; ++c) {
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
// dV accumulates in lane-private LDS words per wave; the waves of a workgroup are summed in fixed order into
// one partial row, like the cross network's parameter gradients.
template <int KT>
__global__ __launch_bounds__(256) void nfm_bi_bwd_kernel(const float* __restrict__ E, const float* __restrict__ V,
                                                         const float* __restrict__ s_in, const float* __restrict__ dout,
                                                         int B, int D, int K, float* __restrict__ dE,
                                                         float* __restrict__ partials) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float* acc = sm + wv * D * K;     // [D][K], row n private to lane n % 64
    for (int i = lane; i < D; i += 64)
        for (int c = 0; c < K; ++c) acc[i * K + c] = 0.f;
    for (int b = blockIdx.x * nw + wv; b < B; b += gridDim.x * nw) {
        float g[KT], s[KT];
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            g[c] = c < K ? dout[(long long)b * K + c] : 0.f;
            s[c] = c < K ? s_in[(long long)b * K + c] : 0.f;
        }
        for (int n = lane; n < D; n += 64) {
            const float e = E[(long long)b * D + n];
            float de = 0.f;
#pragma unroll
            for (int c = 0; c < KT; ++c) {
                if (c < K) {
                    const float v = V[n * K + c];
                    de = fmaf(g[c], s[c] * v - e * v * v, de);
                    acc[n * K + c] = fmaf(g[c] * e, s[c] - e * v, acc[n * K + c]);
                }
            }
            dE[(long long)b * D + n] = de;
        }
    }
    __syncthreads();
    float* prow = partials + (long long)blockIdx.x * D * K;
    for (int i = threadIdx.x; i < D * K; i += blockDim.x) {
        float t = 0.f;
        for (int v = 0; v < nw; ++v) t += sm[v * D * K + i];
        prow[i] = t;
    }
}

// Register forms of the two NFM kernels for D <= 64*T: V sits in LDS once per workgroup, a lane's T elements of a
// sample are loaded up front, and (backward) its T*KT dV accumulators stay in registers.  Same arithmetic and
// summation orders as the kernels above.
// Row n of the LDS copy of V: 16-byte reads when the row is exactly KT (>= 4) floats, else scalar reads.
template <int KT>
__device__ __forceinline__ void nfm_load_v(const float* sm, int n, int K, float (&vr)[KT]) {
    if constexpr (KT >= 4) {
        if (K == KT) {
#pragma unroll
            for (int c4 = 0; c4 < KT / 4; ++c4) {
                const float4 t = reinterpret_cast<const float4*>(sm + n * KT)[c4];
                vr[4 * c4] = t.x; vr[4 * c4 + 1] = t.y; vr[4 * c4 + 2] = t.z; vr[4 * c4 + 3] = t.w;
            }
            return;
        }
    }
#pragma unroll
    for (int c = 0; c < KT; ++c) vr[c] = c < K ? sm[n * K + c] : 0.f;
}

template <int T, int KT>
__global__ __launch_bounds__(256) void nfm_bi_fwd_reg_kernel(const float* __restrict__ E, const float* __restrict__ V, int B,
                                                             int D, int K, float* __restrict__ out,
                                                             float* __restrict__ s_out) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < D * K; i += 256) sm[i] = V[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int b = blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += gridDim.x * 4) {
        float e[T];
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int n = lane + 64 * i;
            e[i] = n < D ? E[(long long)b * D + n] : 0.f;
        }
        float s[KT], q[KT];
#pragma unroll
        for (int c = 0; c < KT; ++c) s[c] = q[c] = 0.f;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int n = lane + 64 * i;
            if (n < D) {
                float vr[KT];
                nfm_load_v<KT>(sm, n, K, vr);
#pragma unroll
                for (int c = 0; c < KT; ++c) {
                    if (c < K) {
                        s[c] = fmaf(e[i], vr[c], s[c]);
                        q[c] = fmaf(e[i] * e[i], vr[c] * vr[c], q[c]);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            if (c < K) {
                const float sc = rs_wave_sum(s[c]), qc = rs_wave_sum(q[c]);
                if (lane == 0) {
                    out[(long long)b * K + c] = 0.5f * (sc * sc - qc);
                    if (s_out) s_out[(long long)b * K + c] = sc;
                }
            }
        }
    }
}

template <int T, int KT>
__global__ __launch_bounds__(256) void nfm_bi_bwd_reg_kernel(const float* __restrict__ E, const float* __restrict__ V,
                                                             const float* __restrict__ s_in, const float* __restrict__ dout,
                                                             int B, int D, int K, float* __restrict__ dE,
                                                             float* __restrict__ partials) {
    extern __shared__ float sm[];        // first V, at the end the four waves' accumulators [4][D*K]
    for (int i = threadIdx.x; i < D * K; i += 256) sm[i] = V[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc[T][KT];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int c = 0; c < KT; ++c) acc[i][c] = 0.f;
    for (int b = blockIdx.x * 4 + wv; b < B; b += gridDim.x * 4) {
        float g[KT], s[KT], e[T];
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            g[c] = c < K ? dout[(long long)b * K + c] : 0.f;
            s[c] = c < K ? s_in[(long long)b * K + c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int n = lane + 64 * i;
            e[i] = n < D ? E[(long long)b * D + n] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int n = lane + 64 * i;
            if (n < D) {
                float de = 0.f;
                float vr[KT];
                nfm_load_v<KT>(sm, n, K, vr);
#pragma unroll
                for (int c = 0; c < KT; ++c) {
                    if (c < K) {
                        const float v = vr[c];
                        de = fmaf(g[c], s[c] * v - e[i] * v * v, de);
                        acc[i][c] = fmaf(g[c] * e[i], s[c] - e[i] * v, acc[i][c]);
                    }
                }
                dE[(long long)b * D + n] = de;
            }
        }
    }
    __syncthreads();                     // every wave is done with V
    float* mine = sm + wv * D * K;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int n = lane + 64 * i;
        if (n < D) {
#pragma unroll
            for (int c = 0; c < KT; ++c)
                if (c < K) mine[n * K + c] = acc[i][c];
        }
    }
    __syncthreads();
    float* prow = partials + (long long)blockIdx.x * D * K;
    for (int i = threadIdx.x; i < D * K; i += 256)
        prow[i] = ((sm[i] + sm[D * K + i]) + sm[2 * D * K + i]) + sm[3 * D * K + i];
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


// Wave-per-sample form of the two AFM kernels above for F <= 64 fields.  Lane j owns field j (its embedding stays
// in registers); in round t = 1..F/2 it meets partner (j + t) mod F, so every unordered pair is visited exactly
// once (for even F the last round is taken by the lanes j < F/2 only) and every field is the partner of exactly
// one lane per round.  That makes the backward deterministic without atomics: what a pair contributes to its
// partner's gradient travels there by one cross-lane read per round.  Partner rows are read from an LDS copy of
// the sample with row stride D+1 (conflict-free); the attention parameters are wave-uniform and come through the
// scalar data path.  Work per pair is ~2*A*D FMAs forward and ~6*A*D backward: these kernels are bound by the
// vector ALUs, not by HBM.
__device__ __forceinline__ bool afm_active(int j, int t, int F) { return j < F && (2 * t != F || 2 * j < F); }

template <int D>
__device__ __forceinline__ void afm_load_row(const float* __restrict__ src, bool valid, float* ej) {
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v = reinterpret_cast<const float4*>(src)[c];
        ej[4 * c] = v.x; ej[4 * c + 1] = v.y; ej[4 * c + 2] = v.z; ej[4 * c + 3] = v.w;
    }
}

// Scores of the lane's pairs: s -> sc[t-1][lane] (-inf where the lane has no pair in round t), pw . p -> gr[t-1][lane];
// returns the lane's maximum score.
template <int D, int A>
__device__ __forceinline__ float afm_score_pass(const float* ej, const float* e, float* sc, float* gr, int lane, int F,
                                                const float* __restrict__ W, const float* __restrict__ bA,
                                                const float* __restrict__ h, const float* __restrict__ pw) {
    constexpr int ES = D + 1;
    const int T = F / 2;
    float m = -INFINITY;
    for (int t = 1; t <= T; ++t) {
        int pr = lane + t;
        if (pr >= F) pr -= F;
        float s = -INFINITY, gsum = 0.f;
        if (lane < F) {
            float z[A];
#pragma unroll
            for (int a = 0; a < A; ++a) z[a] = bA[a];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const float pk = ej[k] * e[pr * ES + k];
                gsum = fmaf(pw[k], pk, gsum);
#pragma unroll
                for (int a = 0; a < A; ++a) z[a] = fmaf(W[a * D + k], pk, z[a]);
            }
            if (afm_active(lane, t, F)) {
                s = 0.f;
#pragma unroll
                for (int a = 0; a < A; ++a) s = fmaf(h[a], fmaxf(z[a], 0.f), s);
            }
        }
        sc[(t - 1) * 64 + lane] = s;
        gr[(t - 1) * 64 + lane] = gsum;
        m = fmaxf(m, s);
    }
    return m;
}

template <int D, int A>
__global__ __launch_bounds__(256) void afm_fwd_wave_kernel(const float* __restrict__ emb, int B, int F,
                                                           const float* __restrict__ W, const float* __restrict__ bA,
                                                           const float* __restrict__ h, const float* __restrict__ pw,
                                                           const float* __restrict__ pb, float* __restrict__ out,
                                                           int region) {
    extern __shared__ float sm[];
    constexpr int ES = D + 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int T = F / 2;
    float* e = sm + wv * region;
    float* sc = e + F * ES;
    float* gr = sc + T * 64;
    for (int b = blockIdx.x * 4 + wv; b < B; b += gridDim.x * 4) {
        float ej[D];
        afm_load_row<D>(emb + ((long long)b * F + lane) * D, lane < F, ej);
        __builtin_amdgcn_wave_barrier();
        if (lane < F) {
#pragma unroll
            for (int k = 0; k < D; ++k) e[lane * ES + k] = ej[k];
        }
        __builtin_amdgcn_wave_barrier();
        float m = afm_score_pass<D, A>(ej, e, sc, gr, lane, F, W, bA, h, pw);
        m = wave_max(m);
        float sum = 0.f, o = 0.f;
        for (int t = 0; t < T; ++t) {
            const float ex = expf(sc[t * 64 + lane] - m);
            sum += ex;
            o = fmaf(ex, gr[t * 64 + lane], o);
        }
        sum = rs_wave_sum(sum);
        o = rs_wave_sum(o);
        if (lane == 0) out[b] = o / sum + pb[0];
    }
}

template <int D, int A>
__global__ __launch_bounds__(256) void afm_bwd_wave_kernel(const float* __restrict__ emb, const float* __restrict__ dout,
                                                           int B, int F, const float* __restrict__ W,
                                                           const float* __restrict__ bA, const float* __restrict__ h,
                                                           const float* __restrict__ pw, float* __restrict__ demb,
                                                           float* __restrict__ partials, int region) {
    extern __shared__ float sm[];
    constexpr int ES = D + 1;
    constexpr int ZS = A + 1, PS = D + 1;        // row strides of the per-pair staging (bank-conflict padding)
    constexpr int NB = D > 16 ? D / 16 : 1;      // 16-column blocks of dW
    constexpr int NP = A * D + 2 * A + D + 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int T = F / 2;
    float* e = sm + wv * region;
    float* sc = e + F * ES;
    float* gr = sc + T * 64;
    float* zs = gr + T * 64;                     // [64][ZS]  dL/dz of the lane's pair in this round
    float* ps = zs + 64 * ZS;                    // [64][PS]  p = e_i * e_j of that pair
    for (int i = lane; i < 64 * (ZS + PS); i += 64) zs[i] = 0.f;      // lanes >= F never write their rows
    // dW[a][k] = sum over pairs dz_a * p_k: accumulated over rounds and samples as a [16 x 16*NB] MFMA tile whose
    // summation index runs over the wave's 64 pair slots (4 per v_mfma_f32_16x16x4_f32)
    f32x4 accW[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) accW[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float accb[A], acch[A], accpw[D], accpb = 0.f;
#pragma unroll
    for (int i = 0; i < A; ++i) accb[i] = acch[i] = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) accpw[i] = 0.f;
    const int ksteps = (F + 3) / 4;

    for (int b = blockIdx.x * 4 + wv; b < B; b += gridDim.x * 4) {
        float ej[D];
        afm_load_row<D>(emb + ((long long)b * F + lane) * D, lane < F, ej);
        __builtin_amdgcn_wave_barrier();
        if (lane < F) {
#pragma unroll
            for (int k = 0; k < D; ++k) e[lane * ES + k] = ej[k];
        }
        __builtin_amdgcn_wave_barrier();
        const float go = dout[b];
        float m = afm_score_pass<D, A>(ej, e, sc, gr, lane, F, W, bA, h, pw);
        m = wave_max(m);
        float sum = 0.f;
        for (int t = 0; t < T; ++t) {
            const float ex = expf(sc[t * 64 + lane] - m);
            sc[t * 64 + lane] = ex;
            sum += ex;
        }
        const float inv = 1.f / rs_wave_sum(sum);
        float gbar = 0.f;
        for (int t = 0; t < T; ++t) {
            const float a = sc[t * 64 + lane] * inv;
            const float g = go * gr[t * 64 + lane];
            sc[t * 64 + lane] = a;
            gr[t * 64 + lane] = g;
            gbar = fmaf(a, g, gbar);
        }
        gbar = rs_wave_sum(gbar);
        if (lane == 0) accpb += go;

        float de[D];
#pragma unroll
        for (int k = 0; k < D; ++k) de[k] = 0.f;
        for (int t = 1; t <= T; ++t) {
            int pr = lane + t;
            if (pr >= F) pr -= F;
            int src = lane - t;
            if (src < 0) src += F;
            const float a = sc[(t - 1) * 64 + lane];               // 0 where the lane has no pair in this round
            const float ds = a * (gr[(t - 1) * 64 + lane] - gbar);
            float send[D];
#pragma unroll
            for (int k = 0; k < D; ++k) send[k] = 0.f;
            __builtin_amdgcn_wave_barrier();                       // last round's staging has been consumed
            if (lane < F) {
                float ep[D], pk[D], z[A], dp[D];
#pragma unroll
                for (int a_ = 0; a_ < A; ++a_) z[a_] = bA[a_];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    ep[k] = e[pr * ES + k];
                    pk[k] = ej[k] * ep[k];
                    ps[lane * PS + k] = pk[k];
                    dp[k] = a * go * pw[k];
                    accpw[k] = fmaf(go * a, pk[k], accpw[k]);
#pragma unroll
                    for (int a_ = 0; a_ < A; ++a_) z[a_] = fmaf(W[a_ * D + k], pk[k], z[a_]);
                }
#pragma unroll
                for (int a_ = 0; a_ < A; ++a_) {
                    const bool on = z[a_] > 0.f;
                    const float dz = on ? ds * h[a_] : 0.f;
                    zs[lane * ZS + a_] = dz;
                    accb[a_] += dz;
                    acch[a_] = fmaf(ds, on ? z[a_] : 0.f, acch[a_]);
#pragma unroll
                    for (int k = 0; k < D; ++k) dp[k] = fmaf(dz, W[a_ * D + k], dp[k]);
                }
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    de[k] = fmaf(dp[k], ep[k], de[k]);
                    send[k] = dp[k] * ej[k];
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int s4 = 0; s4 < ksteps; ++s4) {
                const int pl = 4 * s4 + g4;
                const float av = r16 < A ? zs[pl * ZS + r16] : 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float bv = 16 * nb + r16 < D ? ps[pl * PS + 16 * nb + r16] : 0.f;
                    accW[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accW[nb], 0, 0, 0);
                }
            }
            const bool take = lane < F && afm_active(src, t, F);
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const float r = __shfl(send[k], src, 64);
                if (take) de[k] += r;
            }
        }
        if (lane < F) {
            float4* dst = reinterpret_cast<float4*>(demb + ((long long)b * F + lane) * D);
#pragma unroll
            for (int c = 0; c < D / 4; ++c) dst[c] = make_float4(de[4 * c], de[4 * c + 1], de[4 * c + 2], de[4 * c + 3]);
        }
    }
    // per-wave results -> one row per workgroup (waves added in fixed order)
    __builtin_amdgcn_wave_barrier();
    float* red = sm + wv * region;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int a_ = 4 * g4 + v, k = 16 * nb + r16;
            if (a_ < A && k < D) red[a_ * D + k] = accW[nb][v];
        }
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const float tb = rs_wave_sum(accb[i]), th = rs_wave_sum(acch[i]);
        if (lane == 0) { red[A * D + i] = tb; red[A * D + A + i] = th; }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float t = rs_wave_sum(accpw[i]);
        if (lane == 0) red[A * D + 2 * A + i] = t;
    }
    const float tpb = rs_wave_sum(accpb);
    if (lane == 0) red[A * D + 2 * A + D] = tpb;
    __syncthreads();
    float* prow = partials + (long long)blockIdx.x * NP;
    for (int i = threadIdx.x; i < NP; i += 256) prow[i] = ((sm[i] + sm[region + i]) + sm[2 * region + i]) + sm[3 * region + i];
}

// ---- matrix-core form of the two wave kernels for d <= 16 ------------------------------------------------------
// Same round-robin schedule (pair slot q = field q with partner (q + t) mod F in round t), but the work is laid
// out as v_mfma_f32_16x16x4_f32 tiles of 16 pair slots and a lane never holds a whole pair: lane l works on pair
// slot 16*tile + l%16 and on the four embedding dims 4*(l/16) .. +3.  With Wx = [W ; pw] (the projection vector as
// one more row, so pw.p and dL/dpw ride along):
//   Z^T  [rows x 16] = Wx . P^T              pre-activations (+ pw.p in row A); P is formed in registers from the
//                                            lane's own four dims of field q and of its partner (one ds_read_b128),
//   dP^T [d x 16]    = Wx^T . dZ^T           the Z^T tile a lane holds IS the B operand of this product (register v
//                                            carries row 4*(l/16) + v, and Wx^T is kept in the matching order); row A
//                                            of dZ^T is a*dout, which adds the a*dout*pw term of dL/dp,
//   dWx  [rows x d] += dZ^T . P              summed over pair slots: the one product whose summation index is the
//                                            pair, so dZ^T and P pass through LDS once, transposed ([row][slot]),
// and the owner-side embedding gradient accumulates in registers; the partner-side one is added into an LDS copy of
// the sample's gradient (every field is the partner of exactly one slot per round: no two lanes of an instruction
// meet).  The softmax over all pairs is taken online in the first pass (running maximum, rescaled sums), so the
// scores are not stored: the second pass recomputes them from the same instructions, bit for bit.
template <int D, int A>
struct AfmMx {
    static constexpr int AE = A + 1;                      // attention units + the projection row
    static constexpr int RT = (AE + 15) / 16;             // 16-row tiles of Wx
    static constexpr int GRT = A / 16, GA = (A % 16) / 4, GV = A % 4;   // row A: row tile, lane group, register
    static constexpr int KG = D / 4;                      // lane groups that hold embedding dims
    float wz[RT][4];   // A operand of Z^T, step s:   Wx[16rt + l%16][4*(l/16) + s]
    float wd[RT][4];   // A operand of dP^T, step v:  Wx[16rt + 4*(l/16) + v][l%16]
    float hv[RT][4];   // h[a] for a = 16rt + 4*(l/16) + v (0 on row A and below)
    float bz[RT][4];   // bA[a], likewise
};

// max(x, 0) in one instruction (fmaxf first quiets a signalling NaN, which an MFMA result never is).
__device__ __forceinline__ float afm_relu(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

template <int D, int A>
__device__ __forceinline__ float afm_wx(const float* __restrict__ W, const float* __restrict__ pw, int a, int k) {
    if (k >= D) return 0.f;
    return a < A ? W[a * D + k] : (a == A ? pw[k] : 0.f);
}

template <int D, int A>
__device__ __forceinline__ void afm_mx_load(AfmMx<D, A>& o, const float* __restrict__ W, const float* __restrict__ bA,
                                            const float* __restrict__ h, const float* __restrict__ pw, int r16, int g4) {
#pragma unroll
    for (int rt = 0; rt < AfmMx<D, A>::RT; ++rt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            o.wz[rt][s] = afm_wx<D, A>(W, pw, 16 * rt + r16, 4 * g4 + s);
            const int a = 16 * rt + 4 * g4 + s;
            o.wd[rt][s] = afm_wx<D, A>(W, pw, a, r16);
            o.hv[rt][s] = a < A ? h[a] : 0.f;
            o.bz[rt][s] = a < A ? bA[a] : 0.f;
        }
    }
}

// One round over all NT tiles, stage by stage so that the tiles' dependency chains interleave: partner dims ->
// part, p = own * part, Z^T tiles -> c, scores s (every lane) and pw.p (-> g, meaningful in the lanes of group GA).
// j4[tile] = 4 * partner field of the tile's slot, advanced here by one field per round (wrapping at F; slots beyond
// F stay on the spare slot F, whose dims are zero).  valid[tile]: the lane's pair slot holds a pair in this round.
template <int D, int A, int NT>
__device__ __forceinline__ void afm_mx_round(const AfmMx<D, A>& o, const float* e4, int GS, int F, int t, int r16, int g4,
                                             const float (&own)[NT][4], float (&part)[NT][4], float (&p)[NT][4],
                                             f32x4 (&c)[NT][AfmMx<D, A>::RT], float (&s)[NT], float (&g)[NT],
                                             int (&j4)[NT], bool (&valid)[NT]) {
    using M = AfmMx<D, A>;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const int f = 16 * tt + r16;
        valid[tt] = f < F && (2 * t != F || 2 * f < F);
        j4[tt] += f < F ? 4 : 0;
        if (j4[tt] >= 4 * F && f < F) j4[tt] -= 4 * F;
        const float4 pv = *reinterpret_cast<const float4*>(e4 + g4 * GS + j4[tt]);
        part[tt][0] = pv.x; part[tt][1] = pv.y; part[tt][2] = pv.z; part[tt][3] = pv.w;
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int k = 0; k < 4; ++k) p[tt][k] = own[tt][k] * part[tt][k];
#pragma unroll
    for (int rt = 0; rt < M::RT; ++rt) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) c[tt][rt] = f32x4{o.bz[rt][0], o.bz[rt][1], o.bz[rt][2], o.bz[rt][3]};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
                c[tt][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.wz[rt][k], p[tt][k], c[tt][rt], 0, 0, 0);
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        float partial = 0.f;
#pragma unroll
        for (int rt = 0; rt < M::RT; ++rt)
#pragma unroll
            for (int v = 0; v < 4; ++v) partial = fmaf(o.hv[rt][v], afm_relu(c[tt][rt][v]), partial);
        s[tt] = partial;
        g[tt] = c[tt][M::GRT][M::GV];
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) s[tt] += __shfl_xor(s[tt], 16, 64);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) s[tt] += __shfl_xor(s[tt], 32, 64);
}

// 4 * partner field before the first round: the slot's own field (the spare slot F for slots beyond F).
template <int NT>
__device__ __forceinline__ void afm_mx_start(int (&j4)[NT], int F, int r16) {
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) j4[tt] = 16 * tt + r16 < F ? 4 * (16 * tt + r16) : 4 * F;
}

// First pass over a sample: online softmax statistics of the lane's pair column (lanes of group GA), then the
// wave-wide maximum, sum of exponentials and sum of exponentials x pw.p.
template <int D, int A, int NT>
__device__ __forceinline__ void afm_mx_pass1(const AfmMx<D, A>& o, const float* e4, int GS, int F, int r16, int g4,
                                             const float (&own)[NT][4], float& mw, float& sumw, float& accw) {
    using M = AfmMx<D, A>;
    const int T = F / 2;
    float m = -INFINITY, sum = 0.f, acc = 0.f;
    int j4[NT];
    afm_mx_start<NT>(j4, F, r16);
    for (int t = 1; t <= T; ++t) {
        float part[NT][4], p[NT][4], s[NT], g[NT];
        f32x4 c[NT][M::RT];
        bool valid[NT];
        afm_mx_round<D, A, NT>(o, e4, GS, F, t, r16, g4, own, part, p, c, s, g, j4, valid);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            // running maximum: exactly one of the two rescaling factors differs from 1, so one exponential serves
            const float dlt = s[tt] - m;                   // +inf while m is still -inf
            const float e1 = __expf(-fabsf(dlt));
            const bool up = valid[tt] && dlt > 0.f;
            const float keep = up ? e1 : 1.f;
            const float ex = valid[tt] ? (up ? 1.f : e1) : 0.f;
            sum = fmaf(sum, keep, ex);
            acc = fmaf(acc, keep, ex * g[tt]);
            m = up ? s[tt] : m;
        }
    }
    const bool mine = g4 == M::GA && m != -INFINITY;
    mw = wave_max(mine ? m : -INFINITY);
    const float w = mine ? __expf(m - mw) : 0.f;
    sumw = rs_wave_sum(sum * w);
    accw = rs_wave_sum(acc * w);
}

// Sample b in tile layout: own[tile] = the lane's four dims of field 16*tile + l%16 (zero beyond F and beyond d); the
// same values go to the LDS copy e4 [4 dim groups][F + 1][4] that partner reads come from (slot F: zeros, the partner
// of the pair slots beyond F), and the partner-side gradient de4 of the same shape is cleared (its slot F takes the
// stores of lanes without a pair).
template <int D, int A, int NT>
__device__ __forceinline__ void afm_mx_sample(const float* __restrict__ emb, long long b, int F, int GS, int r16, int g4,
                                              float* e4, float* de4, float (&own)[NT][4]) {
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const int f = 16 * tt + r16;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < F && 4 * g4 < D) v = *reinterpret_cast<const float4*>(emb + (b * F + f) * D + 4 * g4);
        own[tt][0] = v.x; own[tt][1] = v.y; own[tt][2] = v.z; own[tt][3] = v.w;
        if (f <= F) {
            *reinterpret_cast<float4*>(e4 + g4 * GS + 4 * f) = v;
            if (de4) *reinterpret_cast<float4*>(de4 + g4 * GS + 4 * f) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

template <int D, int A, int NT>
__global__ __launch_bounds__(256) void afm_fwd_mfma_kernel(const float* __restrict__ emb, int B, int F,
                                                           const float* __restrict__ W, const float* __restrict__ bA,
                                                           const float* __restrict__ h, const float* __restrict__ pw,
                                                           const float* __restrict__ pb, float* __restrict__ out,
                                                           int region) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int GS = 4 * (F + 1);
    float* e4 = sm + wv * region;
    AfmMx<D, A> o;
    afm_mx_load<D, A>(o, W, bA, h, pw, r16, g4);
    for (int b = blockIdx.x * 4 + wv; b < B; b += gridDim.x * 4) {
        float own[NT][4];
        __builtin_amdgcn_wave_barrier();                   // the previous sample's partner reads are done
        afm_mx_sample<D, A, NT>(emb, b, F, GS, r16, g4, e4, nullptr, own);
        __builtin_amdgcn_wave_barrier();
        float mw, sumw, accw;
        afm_mx_pass1<D, A, NT>(o, e4, GS, F, r16, g4, own, mw, sumw, accw);
        if (lane == 0) out[b] = accw / sumw + pb[0];
    }
}

template <int D, int A, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void afm_bwd_mfma_kernel(
    const float* __restrict__ emb, const float* __restrict__ dout, int B, int F, const float* __restrict__ W,
    const float* __restrict__ bA, const float* __restrict__ h, const float* __restrict__ pw, float* __restrict__ demb,
    float* __restrict__ partials, int region) {
    using M = AfmMx<D, A>;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NP = A * D + 2 * A + D + 1;
    constexpr int RS = 16 * NT + 4;                       // row stride of the [row][slot] staging (52 / 68: no bank repeats over 16 rows)
    constexpr int NS = 4 * NT;                            // k-steps of the dWx product; lane group g holds slots NS*g .. NS*g + NS-1
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int T = F / 2, GS = 4 * (F + 1);
    float* e4 = sm + wv * region;                         // [4][F+1][4]   the sample
    float* de4 = e4 + 4 * GS;                             // [4][F+1][4]   partner-side gradient
    float* zsT = de4 + 4 * GS;                            // [16*RT][RS]   dZ^T rows by pair slot
    float* psT = zsT + 16 * M::RT * RS;                   // [16][RS]      P^T rows by pair slot
    AfmMx<D, A> o;
    afm_mx_load<D, A>(o, W, bA, h, pw, r16, g4);
    f32x4 accW[M::RT];                                    // dWx[16rt + 4*g4 + v][r16]
    float accb[M::RT][4], acch[M::RT][4], accpb = 0.f;    // row 16rt + 4*g4 + v, summed over the lane's pair column
#pragma unroll
    for (int rt = 0; rt < M::RT; ++rt) {
        accW[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 4; ++v) accb[rt][v] = acch[rt][v] = 0.f;
    }

    for (int b = blockIdx.x * 4 + wv; b < B; b += gridDim.x * 4) {
        float own[NT][4];
        __builtin_amdgcn_wave_barrier();                   // the previous sample's LDS reads are done
        afm_mx_sample<D, A, NT>(emb, b, F, GS, r16, g4, e4, de4, own);
        __builtin_amdgcn_wave_barrier();
        const float go = dout[b];
        float mw, sumw, accw;
        afm_mx_pass1<D, A, NT>(o, e4, GS, F, r16, g4, own, mw, sumw, accw);
        const float inv = 1.f / sumw;
        const float gbar = go * accw * inv;
        if (lane == 0) accpb += go;

        float de[NT][4];
        int j4[NT];
        afm_mx_start<NT>(j4, F, r16);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int k = 0; k < 4; ++k) de[tt][k] = 0.f;
        // every store of the round below is unconditional (rows and slots without data receive zeros, lanes without
        // a pair aim at the spare slot F), so the round is one basic block and its tiles interleave
        for (int t = 1; t <= T; ++t) {
            float part[NT][4], p[NT][4], s[NT], g[NT], ago[NT], ds[NT];
            f32x4 c[NT][M::RT], q[NT];
            bool valid[NT];
            __builtin_amdgcn_wave_barrier();               // last round's staging has been consumed
            afm_mx_round<D, A, NT>(o, e4, GS, F, t, r16, g4, own, part, p, c, s, g, j4, valid);
            // attention weight and dL/ds of the pair, in the lanes that hold pw.p; dL/ds then goes to every lane
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const float a = valid[tt] ? __expf(s[tt] - mw) * inv : 0.f;
                ago[tt] = a * go;
                ds[tt] = a * (go * g[tt] - gbar);
            }
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) ds[tt] = __shfl(ds[tt], 16 * M::GA + r16, 64);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                q[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) psT[(4 * g4 + k) * RS + 16 * tt + r16] = p[tt][k];
            }
#pragma unroll
            for (int rt = 0; rt < M::RT; ++rt) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        const bool on = c[tt][rt][v] > 0.f;
                        float dz = on ? ds[tt] * o.hv[rt][v] : 0.f;
                        accb[rt][v] += dz;
                        acch[rt][v] = fmaf(ds[tt], afm_relu(c[tt][rt][v]), acch[rt][v]);
                        if (rt == M::GRT && v == M::GV) dz = g4 == M::GA ? ago[tt] : dz;     // row A of dZ^T
                        zsT[(16 * rt + 4 * g4 + v) * RS + 16 * tt + r16] = dz;
                        q[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.wd[rt][v], dz, q[tt], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // dWx += dZ^T . P over this round's pair slots
#pragma unroll
            for (int rt = 0; rt < M::RT; ++rt) {
                float av[NS], bv[NS];
#pragma unroll
                for (int c4 = 0; c4 < NT; ++c4) {
                    const float4 x = *reinterpret_cast<const float4*>(zsT + (16 * rt + r16) * RS + NS * g4 + 4 * c4);
                    const float4 y = *reinterpret_cast<const float4*>(psT + r16 * RS + NS * g4 + 4 * c4);
                    av[4 * c4] = x.x; av[4 * c4 + 1] = x.y; av[4 * c4 + 2] = x.z; av[4 * c4 + 3] = x.w;
                    bv[4 * c4] = y.x; bv[4 * c4 + 1] = y.y; bv[4 * c4 + 2] = y.z; bv[4 * c4 + 3] = y.w;
                }
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) accW[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bv[s4], accW[rt], 0, 0, 0);
            }
            // q = dL/dp of the pair, dims 4*g4 + v.  Owner side into registers; partner side into the LDS copy: every
            // field is the partner of at most one slot per round, so the read-modify-writes of a round never meet
            // (reads first, stores after)
            float4 cur[NT];
            int tgt[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                tgt[tt] = g4 * GS + (valid[tt] ? j4[tt] : 4 * F);
                cur[tt] = *reinterpret_cast<const float4*>(de4 + tgt[tt]);
            }
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int k = 0; k < 4; ++k) de[tt][k] = fmaf(q[tt][k], part[tt][k], de[tt][k]);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
                *reinterpret_cast<float4*>(de4 + tgt[tt]) =
                    make_float4(fmaf(q[tt][0], own[tt][0], cur[tt].x), fmaf(q[tt][1], own[tt][1], cur[tt].y),
                                fmaf(q[tt][2], own[tt][2], cur[tt].z), fmaf(q[tt][3], own[tt][3], cur[tt].w));
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int f = 16 * tt + r16;
            if (f < F && 4 * g4 < D) {
                const float4 pg = *reinterpret_cast<const float4*>(de4 + g4 * GS + 4 * f);
                *reinterpret_cast<float4*>(demb + ((long long)b * F + f) * D + 4 * g4) =
                    make_float4(de[tt][0] + pg.x, de[tt][1] + pg.y, de[tt][2] + pg.z, de[tt][3] + pg.w);
            }
        }
    }
    // per-wave results -> one row per workgroup (waves added in fixed order)
    __builtin_amdgcn_wave_barrier();
    float* red = sm + wv * region;
#pragma unroll
    for (int rt = 0; rt < M::RT; ++rt) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int a_ = 16 * rt + 4 * g4 + v;
            if (r16 < D) {
                if (a_ < A) red[a_ * D + r16] = accW[rt][v];
                else if (a_ == A) red[A * D + 2 * A + r16] = accW[rt][v];          // row A of dWx = dL/dpw
            }
            float tb = accb[rt][v], th = acch[rt][v];
#pragma unroll
            for (int o_ = 1; o_ < 16; o_ <<= 1) {
                tb += __shfl_xor(tb, o_, 64);
                th += __shfl_xor(th, o_, 64);
            }
            if (a_ < A && r16 == 0) { red[A * D + a_] = tb; red[A * D + A + a_] = th; }
        }
    }
    const float tpb = rs_wave_sum(accpb);
    if (lane == 0) red[A * D + 2 * A + D] = tpb;
    __syncthreads();
    float* prow = partials + (long long)blockIdx.x * NP;
    for (int i = threadIdx.x; i < NP; i += 256) prow[i] = ((sm[i] + sm[region + i]) + sm[2 * region + i]) + sm[3 * region + i];
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

// Vector form for d = 4*LP (LP a power of two <= 16): LP lanes share a pair and read its two d-vectors as
// float4s, so every 4*d-byte block of a table row is fetched by one contiguous request; 64/LP pairs are in flight
// per wave and step.  The (i, j) of every pair sit in an LDS table built once per workgroup.
template <int LP>
__global__ __launch_bounds__(256) void ffm_fwd_vec_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ row_offset,
                                                          const int64_t* __restrict__ vocab, const float* __restrict__ T,
                                                          int B, int F, float* __restrict__ out, int* __restrict__ err_flag) {
    extern __shared__ float sm[];
    constexpr int d = 4 * LP, PPW = 64 / LP;
    const int P = F * (F - 1) / 2, Ppad = (P + 1) & ~1;
    int* tab = reinterpret_cast<int*>(sm);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long* rows = reinterpret_cast<long long*>(tab + Ppad) + wv * F;
    for (int p = threadIdx.x; p < P; p += 256) {
        int i, j;
        pair_decode(p, F, &i, &j);
        tab[p] = i | (j << 16);
    }
    __syncthreads();
    const int sub = lane % LP, pl = lane / LP;
    const long long fd = (long long)F * d;
    for (int b = blockIdx.x * 4 + wv; b < B; b += gridDim.x * 4) {
        __builtin_amdgcn_wave_barrier();
        for (int f = lane; f < F; f += 64) {
            long long x = idx[(long long)f * B + b];
            if (x < 0 || x >= vocab[f]) { x = 0; if (err_flag) atomicOr(err_flag, 1); }
            rows[f] = (row_offset[f] + x) * fd;
        }
        __builtin_amdgcn_wave_barrier();
        float acc = 0.f;
#pragma unroll 4
        for (int p0 = 0; p0 < P; p0 += PPW) {
            const int p = p0 + pl;
            if (p < P) {
                const int ij = tab[p], i = ij & 0xffff, j = ij >> 16;
                const float4 x = *reinterpret_cast<const float4*>(T + rows[i] + j * d + 4 * sub);
                const float4 y = *reinterpret_cast<const float4*>(T + rows[j] + i * d + 4 * sub);
                acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
            }
        }
        acc = rs_wave_sum(acc);
        if (lane == 0) out[b] = acc;
    }
}

// FFM on rows that have already been gathered (the row-sharded path): G [B, F, F*d], G[b, i, :] = the field-aware
// row of sample b's value in field i.  out[b] = sum_{i<j} <G[b,i,j*d..], G[b,j,i*d..]>; the gradient with respect to
// the gathered rows is dG[b,i,j*d..] = dout[b] * G[b,j,i*d..] for j != i and zero for the own-field block.
template <int LP>
__global__ __launch_bounds__(256) void ffm_rows_fwd_kernel(const float* __restrict__ G, int B, int F, float* __restrict__ out) {
    extern __shared__ float sm[];
    constexpr int d = 4 * LP, PPW = 64 / LP;
    const int P = F * (F - 1) / 2;
    int* tab = reinterpret_cast<int*>(sm);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int p = threadIdx.x; p < P; p += 256) {
        int i, j;
        pair_decode(p, F, &i, &j);
        tab[p] = i | (j << 16);
    }
    __syncthreads();
    const int sub = lane % LP, pl = lane / LP;
    const long long fd = (long long)F * d;
    for (int b = blockIdx.x * 4 + wv; b < B; b += gridDim.x * 4) {
        const float* g = G + (long long)b * F * fd;
        float acc = 0.f;
#pragma unroll 4
        for (int p0 = 0; p0 < P; p0 += PPW) {
            const int p = p0 + pl;
            if (p < P) {
                const int ij = tab[p], i = ij & 0xffff, j = ij >> 16;
                const float4 x = *reinterpret_cast<const float4*>(g + i * fd + j * d + 4 * sub);
                const float4 y = *reinterpret_cast<const float4*>(g + j * fd + i * d + 4 * sub);
                acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
            }
        }
        acc = rs_wave_sum(acc);
        if (lane == 0) out[b] = acc;
    }
}

__global__ __launch_bounds__(256) void ffm_rows_bwd_kernel(const float* __restrict__ G, const float* __restrict__ dout, int B,
                                                           int F, int d, float* __restrict__ dG) {
    const int fdv = F * d / 4;                 // float4s per row
    const long long total = (long long)B * F * fdv;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const int c4 = (int)(q % fdv);
        const long long bi = q / fdv;
        const int i = (int)(bi % F);
        const long long b = bi / F;
        const int c = 4 * c4, j = c / d, k = c - j * d;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j != i) {
            const float g = dout[b];
            const float4 t = *reinterpret_cast<const float4*>(G + ((b * F + j) * (long long)F + i) * d + k);
            v = make_float4(g * t.x, g * t.y, g * t.z, g * t.w);
        }
        reinterpret_cast<float4*>(dG)[q] = v;
    }
}

// Gradients of the distinct table rows of a batch, compact: rowgrad[f, s, :] (F*d floats) for segment s of field f
// of the sorted batch (see sparse.hip), in the layout the touched-rows optimizer kernels read, plus sum-of-squares
// partials for the global gradient norm.
//   rowgrad[f, s, j*d + k] = sum_b dout[b] * T[r_j(b)][f*d + k]   (j != f; the own-field block stays zero).
// Two phases, both in fixed summation order (no atomics).  Phase 1: one workgroup per chunk of the plan (at most
// CH consecutive entries of one row; a hot row of a skewed batch spans many chunks, which proceed in parallel); the
// sum goes straight to rowgrad when the row has a single chunk, else to chunk_part[f, chunk, :].  Phase 2 adds the
// chunk sums of the multi-chunk rows in chunk order.
constexpr int FFM_TILE = 64;         // >= entries per chunk of a plan built for 1-dim rows
__global__ __launch_bounds__(256) void ffm_bwd_chunks_kernel(PlanView pv, const int64_t* __restrict__ idx,
                                                             const int64_t* __restrict__ row_offset,
                                                             const int64_t* __restrict__ vocab, const float* __restrict__ T,
                                                             const float* __restrict__ dout, int B, int F, int cap, int d,
                                                             float* __restrict__ rowgrad, float* __restrict__ chunk_part,
                                                             float* __restrict__ norm_partials) {
    __shared__ float red[4];
    __shared__ int sb[FFM_TILE];     // samples of the chunk's entries
    __shared__ float sg[FFM_TILE];   // their dL/dout
    constexpr int U = 8;             // independent load chains in flight per thread
    const int f = blockIdx.y;
    const int nchunk = pv.counts[f * 4 + 1];
    const int* sfc = pv.seg_first_chunk + (long long)f * (cap + 1);
    const int* cst = pv.chunk_start + (long long)f * (cap + 1);
    const int* ssrc = pv.sorted_src + (long long)f * cap;
    const int* eseg = pv.entry_seg + (long long)f * cap;
    const long long fd = (long long)F * d;
    // one float4 of the row per thread (host side: F*d <= 1024)
    const int c = 4 * threadIdx.x, j = c / d, k = c - j * d;
    const bool has = c < F * d, live = has && j != f;
    const long long V = live ? vocab[j] : 0, ro = live ? row_offset[j] : 0;
    const int64_t* col = idx + (long long)(live ? j : 0) * B;
    const float* Tf = T + (long long)f * d + k;
    float nsq = 0.f;
    for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
        const int beg = cst[ch], end = cst[ch + 1];
        const int seg = eseg[beg];
        const bool single = sfc[seg + 1] - sfc[seg] == 1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q0 = beg; q0 < end; q0 += FFM_TILE) {
            const int n = end - q0 < FFM_TILE ? end - q0 : FFM_TILE;
            __syncthreads();
            if ((int)threadIdx.x < n) {
                const int b = ssrc[q0 + threadIdx.x] / F;
                sb[threadIdx.x] = b;
                sg[threadIdx.x] = dout[b];
            }
            __syncthreads();
            if (live) {
                for (int t0 = 0; t0 < n; t0 += U) {
                    long long x[U];
                    float4 tv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int t = t0 + u < n ? t0 + u : n - 1;
                        const long long xx = col[sb[t]];
                        x[u] = (xx < 0 || xx >= V) ? 0 : xx;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) tv[u] = *reinterpret_cast<const float4*>(Tf + (ro + x[u]) * fd);
#pragma unroll
                    for (int u = 0; u < U; ++u) {                  // ascending entries: fixed summation order
                        if (t0 + u < n) {
                            const float g = sg[t0 + u];
                            acc.x = fmaf(g, tv[u].x, acc.x); acc.y = fmaf(g, tv[u].y, acc.y);
                            acc.z = fmaf(g, tv[u].z, acc.z); acc.w = fmaf(g, tv[u].w, acc.w);
                        }
                    }
                }
            }
        }
        if (has) {
            float* dst = single ? rowgrad + ((long long)f * cap + seg) * fd : chunk_part + ((long long)f * cap + ch) * fd;
            *reinterpret_cast<float4*>(dst + c) = acc;
            if (single) nsq += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
        }
    }
    const float tot = rs_block_sum<256>(nsq, red);
    if (threadIdx.x == 0) norm_partials[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void ffm_bwd_finish_kernel(PlanView pv, int F, int cap, int d,
                                                             float* __restrict__ rowgrad, const float* __restrict__ chunk_part,
                                                             float* __restrict__ norm_partials) {
    __shared__ float red[4];
    const int f = blockIdx.y;
    const int nseg = pv.counts[f * 4 + 2];
    const int* sfc = pv.seg_first_chunk + (long long)f * (cap + 1);
    const long long fd = (long long)F * d;
    const int c = 4 * threadIdx.x;
    const bool has = c < F * d;
    float nsq = 0.f;
    for (int s = blockIdx.x; s < nseg; s += gridDim.x) {
        const int c0 = sfc[s], c1 = sfc[s + 1];
        if (c1 - c0 < 2 || !has) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ch = c0; ch < c1; ++ch) {
            const float4 t = *reinterpret_cast<const float4*>(chunk_part + ((long long)f * cap + ch) * fd + c);
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        *reinterpret_cast<float4*>(rowgrad + ((long long)f * cap + s) * fd + c) = acc;
        nsq += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
    }
    const float tot = rs_block_sum<256>(nsq, red);
    if (threadIdx.x == 0) norm_partials[blockIdx.y * gridDim.x + blockIdx.x] = tot;
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
constexpr size_t LDS_BIG = 128 * 1024;      // workgroups that ask for more than 64 KB raise the kernel's limit first

// Waves per workgroup (4, 2 or 1) whose private LDS regions of per_wave bytes fit LDS_BIG.
int waves_for(size_t per_wave) { return 4 * per_wave <= LDS_BIG ? 4 : (2 * per_wave <= LDS_BIG ? 2 : 1); }

bool raise_lds(const void* kern) {
    return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BIG) == hipSuccess;
}
constexpr int PARTIAL_ROWS_MAX = 512;
constexpr int FFM_ROW_GROUPS = 1024;   // workgroups per field of rs_ffm_bwd_rows, phase 1 (chunks)
constexpr int FFM_FINISH_GROUPS = 64;  // phase 2 (rows with several chunks)

int flat_blocks(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// Shapes served by the matrix-core PNN kernels (anything else takes the LDS kernels).
bool pnn_mfma_ok(int F, int d, const float* emb) {
    return F <= 64 && (d == 4 || d == 8 || d == 16 || d == 32) && (reinterpret_cast<uintptr_t>(emb) & 15) == 0;
}
int pnn_mfma_grid(int B) { const int g = rs_cdiv(B, 4); return g > 2048 ? 2048 : g; }
#define RS_PNN_CASE(K, NB, DD) case (NB) * 100 + (DD): K(NB, DD); break;
#define RS_PNN_DISPATCH(K)                                                                         \
    switch (rs_cdiv(F, 16) * 100 + d) {                                                            \
        RS_PNN_CASE(K, 1, 4) RS_PNN_CASE(K, 1, 8) RS_PNN_CASE(K, 1, 16) RS_PNN_CASE(K, 1, 32)      \
        RS_PNN_CASE(K, 2, 4) RS_PNN_CASE(K, 2, 8) RS_PNN_CASE(K, 2, 16) RS_PNN_CASE(K, 2, 32)      \
        RS_PNN_CASE(K, 3, 4) RS_PNN_CASE(K, 3, 8) RS_PNN_CASE(K, 3, 16) RS_PNN_CASE(K, 3, 32)      \
        RS_PNN_CASE(K, 4, 4) RS_PNN_CASE(K, 4, 8) RS_PNN_CASE(K, 4, 16) RS_PNN_CASE(K, 4, 32)      \
        default: break;                                                                            \
    }

// The matrix-core AFM kernels with three tiles of 16 pair slots (F <= 48) or four.
#define RS_AFM_NT_ALIAS(NAME, KERN, NTV)                                                                               \
    template <int D, int A>                                                                                            \
    constexpr auto NAME = KERN<D, A, NTV>;
RS_AFM_NT_ALIAS(afm_fwd_mfma3, afm_fwd_mfma_kernel, 3)
RS_AFM_NT_ALIAS(afm_fwd_mfma4, afm_fwd_mfma_kernel, 4)
RS_AFM_NT_ALIAS(afm_bwd_mfma3, afm_bwd_mfma_kernel, 3)
RS_AFM_NT_ALIAS(afm_bwd_mfma4, afm_bwd_mfma_kernel, 4)
#undef RS_AFM_NT_ALIAS

// (embedding_dim, attention_dim) instantiated for the wave-per-sample AFM kernels.
bool afm_wave_shape(int d, int A) {
    return (d == 4 || d == 8 || d == 16 || d == 32) && (A == 4 || A == 8 || A == 16) && A * d <= 256;
}
// Per-wave LDS floats of the vector-ALU wave kernels (d = 32): sample copy [F][d+1], two [F/2][64] pair arrays, the
// backward's per-pair staging [64][A+1 + d+1]; at least one parameter-gradient row.
int afm_region_floats(int F, int d, int A, bool bwd) {
    const int need = F * (d + 1) + 2 * (F / 2) * 64 + (bwd ? 64 * (A + 1 + d + 1) : 0), np = A * d + 2 * A + d + 1;
    return need > np ? need : np;
}
// Per-wave LDS floats of the matrix-core kernels (d <= 16): the sample [4][F + 1][4], and in the backward its
// partner-side gradient, dZ^T [16 * row tiles][16*NT + 4] and P^T [16][16*NT + 4]; at least one parameter-gradient
// row; a multiple of 4 floats (16-byte accesses).
int afm_mx_region_floats(int F, int d, int A, bool bwd) {
    const int nt = F <= 48 ? 3 : 4, rs = 16 * nt + 4, rt = (A + 1 + 15) / 16;
    const int need = 16 * (F + 1) * (bwd ? 2 : 1) + (bwd ? (16 * rt + 16) * rs : 0), np = A * d + 2 * A + d + 1;
    return ((need > np ? need : np) + 3) & ~3;
}
#define RS_AFM_CASE(K, DD, AA) case (DD) * 100 + (AA): K(DD, AA); break;
#define RS_AFM_DISPATCH(KM, KV)                                                                    \
    switch (d * 100 + A) {                                                                         \
        RS_AFM_CASE(KM, 4, 4) RS_AFM_CASE(KM, 4, 8) RS_AFM_CASE(KM, 4, 16)                         \
        RS_AFM_CASE(KM, 8, 4) RS_AFM_CASE(KM, 8, 8) RS_AFM_CASE(KM, 8, 16)                         \
        RS_AFM_CASE(KM, 16, 4) RS_AFM_CASE(KM, 16, 8) RS_AFM_CASE(KM, 16, 16)                      \
        RS_AFM_CASE(KV, 32, 4) RS_AFM_CASE(KV, 32, 8)                                              \
        default: break;                                                                            \
    }

// Elements per lane (T) instantiated for the register forms of the cross-network kernels: smallest >= ceil(D/64).
int dcn_reg_T(int D) {
    const int need = rs_cdiv(D, 64);
    const int opts[] = {2, 4, 7, 10, 13, 16};
    for (int t : opts) if (t >= need) return t;
    return 0;
}
#define RS_DCN_CASE(K, T, L) case (T) * 10 + (L): K(T, L); break;
#define RS_DCN_ROW(K, T) RS_DCN_CASE(K, T, 1) RS_DCN_CASE(K, T, 2) RS_DCN_CASE(K, T, 3) RS_DCN_CASE(K, T, 4)
#define RS_DCN_DISPATCH(K, T, L)                                                                   \
    switch ((T) * 10 + (L)) {                                                                      \
        RS_DCN_ROW(K, 2) RS_DCN_ROW(K, 4) RS_DCN_ROW(K, 7) RS_DCN_ROW(K, 10) RS_DCN_ROW(K, 13) RS_DCN_ROW(K, 16) \
        default: break;                                                                            \
    }

int nfm_KT(int K) { return K <= 1 ? 1 : (K <= 2 ? 2 : (K <= 4 ? 4 : (K <= 8 ? 8 : 16))); }
#define RS_NFM_CASE(K_, T, KT) case (T) * 100 + (KT): K_(T, KT); break;
#define RS_NFM_FWD_ROW(K_, T) RS_NFM_CASE(K_, T, 1) RS_NFM_CASE(K_, T, 2) RS_NFM_CASE(K_, T, 4) RS_NFM_CASE(K_, T, 8) RS_NFM_CASE(K_, T, 16)
#define RS_NFM_FWD_DISPATCH(K_, T, KT)                                                             \
    switch ((T) * 100 + (KT)) {                                                                    \
        RS_NFM_FWD_ROW(K_, 2) RS_NFM_FWD_ROW(K_, 4) RS_NFM_FWD_ROW(K_, 7) RS_NFM_FWD_ROW(K_, 10)   \
        RS_NFM_FWD_ROW(K_, 13) RS_NFM_FWD_ROW(K_, 16)                                              \
        default: break;                                                                            \
    }
// backward: T*KT accumulator registers per lane, instantiated up to 64
#define RS_NFM_BWD_DISPATCH(K_, T, KT)                                                             \
    switch ((T) * 100 + (KT)) {                                                                    \
        RS_NFM_FWD_ROW(K_, 2) RS_NFM_FWD_ROW(K_, 4)                                                \
        RS_NFM_CASE(K_, 7, 1) RS_NFM_CASE(K_, 7, 2) RS_NFM_CASE(K_, 7, 4) RS_NFM_CASE(K_, 7, 8)    \
        RS_NFM_CASE(K_, 10, 1) RS_NFM_CASE(K_, 10, 2) RS_NFM_CASE(K_, 10, 4)                       \
        RS_NFM_CASE(K_, 13, 1) RS_NFM_CASE(K_, 13, 2) RS_NFM_CASE(K_, 13, 4)                       \
        RS_NFM_CASE(K_, 16, 1) RS_NFM_CASE(K_, 16, 2) RS_NFM_CASE(K_, 16, 4)                       \
        default: break;                                                                            \
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
    if (pnn_mfma_ok(F, d, emb)) {
        const int grid = pnn_mfma_grid(B);
#define RS_PNN_F(NB, DD) hipLaunchKernelGGL((pnn_inner_fwd_mfma_kernel<NB, DD, false>), dim3(grid), dim3(256), 0, stream, emb, B, F, out, (long long)(F * (F - 1) / 2))
        RS_PNN_DISPATCH(RS_PNN_F)
#undef RS_PNN_F
        RS_CHECK_LAUNCH("rs_pnn_inner_fwd");
        return RS_OK;
    }
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
    if (pnn_mfma_ok(F, d, emb)) {
        const int grid = pnn_mfma_grid(B);
        const size_t blds = (size_t)4 * (F * (F - 1) / 2 + 1) * sizeof(float);
#define RS_PNN_B(NB, DD) hipLaunchKernelGGL((pnn_inner_bwd_mfma_kernel<NB, DD, false>), dim3(grid), dim3(256), blds, stream, emb, dout, B, F, demb, (long long)(F * (F - 1) / 2))
        RS_PNN_DISPATCH(RS_PNN_B)
#undef RS_PNN_B
        RS_CHECK_LAUNCH("rs_pnn_inner_bwd");
        return RS_OK;
    }
    const size_t lds = (size_t)4 * (F * d + F * (F - 1) / 2) * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_pnn_inner_bwd: F=%d d=%d does not fit the LDS staging", F, d);
    hipLaunchKernelGGL(pnn_inner_bwd_kernel, dim3(rs_cdiv(B, 4)), dim3(256), lds, stream, emb, dout, B, F, d, demb);
    RS_CHECK_LAUNCH("rs_pnn_inner_bwd");
    return RS_OK;
}

// The input of PNN's first MLP layer with pnn_type inner (PNN.py:58): out [B, ld] rows = [emb row (F*d) | inner
// products (P) | zeros], ld >= F*d + P a multiple of four with at most 63 columns of padding; the backward takes the
// gradient of such rows and returns demb = dout[:, :F*d] + the products' share.  1 when the shape is served (the
// matrix-core kernels: F <= 64, d in {4, 8, 16, 32}, 16-byte aligned rows), else 0.
int rs_pnn_inner_cat_ok(int F, int d) { return pnn_mfma_ok(F, d, nullptr) ? 1 : 0; }

int rs_pnn_inner_cat_fwd(const float* emb, int B, int F, int d, float* out, long long ld, hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 1 && d > 0, "rs_pnn_inner_cat_fwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(emb && out, "rs_pnn_inner_cat_fwd: null argument");
    const long long K = (long long)F * d + F * (F - 1) / 2;
    RS_CHECK_ARG(ld >= K && ld % 4 == 0 && ld - K < 64, "rs_pnn_inner_cat_fwd: row stride %lld for %lld columns", ld, K);
    RS_CHECK_ARG(pnn_mfma_ok(F, d, emb) && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "rs_pnn_inner_cat_fwd: F=%d d=%d (or the alignment of emb / out) is not served", F, d);
    const int grid = pnn_mfma_grid(B);
#define RS_PNN_F(NB, DD) hipLaunchKernelGGL((pnn_inner_fwd_mfma_kernel<NB, DD, true>), dim3(grid), dim3(256), 0, stream, emb, B, F, out, ld)
    RS_PNN_DISPATCH(RS_PNN_F)
#undef RS_PNN_F
    RS_CHECK_LAUNCH("rs_pnn_inner_cat_fwd");
    return RS_OK;
}

int rs_pnn_inner_cat_bwd(const float* emb, const float* dout, long long ld, int B, int F, int d, float* demb,
                         hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 1 && d > 0, "rs_pnn_inner_cat_bwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(emb && dout && demb, "rs_pnn_inner_cat_bwd: null argument");
    const long long K = (long long)F * d + F * (F - 1) / 2;
    RS_CHECK_ARG(ld >= K, "rs_pnn_inner_cat_bwd: row stride %lld for %lld columns", ld, K);
    RS_CHECK_ARG(pnn_mfma_ok(F, d, emb), "rs_pnn_inner_cat_bwd: F=%d d=%d (or the alignment of emb) is not served", F, d);
    const int grid = pnn_mfma_grid(B);
    const size_t blds = (size_t)4 * (F * (F - 1) / 2 + 1) * sizeof(float);
#define RS_PNN_B(NB, DD) hipLaunchKernelGGL((pnn_inner_bwd_mfma_kernel<NB, DD, true>), dim3(grid), dim3(256), blds, stream, emb, dout, B, F, demb, ld)
    RS_PNN_DISPATCH(RS_PNN_B)
#undef RS_PNN_B
    RS_CHECK_LAUNCH("rs_pnn_inner_cat_bwd");
    return RS_OK;
}

// PNN outer products: emb [B,F,d] -> out [B, P*d*d] (model/PNN/PNN.py:80-103).
// The *_range forms serve the pairs whose first field lies in [i_lo, i_hi): out / dout hold those pairs only, row stride
// ld floats; the backward adds its result to demb when accumulate is set.
static int pnn_outer_fwd_impl(const float* emb, int B, int F, int d, int i_lo, int i_hi, float* out, long long ld,
                              hipStream_t stream, const char* who) {
    RS_CHECK_ARG(B >= 0 && F > 1 && F < 32768 && d > 0, "%s: bad sizes", who);
    RS_CHECK_ARG(0 <= i_lo && i_lo < i_hi && i_hi <= F, "%s: bad field range [%d, %d)", who, i_lo, i_hi);
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(emb && out, "%s: null argument", who);
    const int P = F * (F - 1) / 2;
    const int p_lo = i_lo * (2 * F - i_lo - 1) / 2, p_hi = i_hi >= F - 1 ? P : i_hi * (2 * F - i_hi - 1) / 2;
    if (p_hi == p_lo) return RS_OK;
    RS_CHECK_ARG((long long)(p_hi - p_lo) * d * d < (1ll << 31) && ld >= (long long)(p_hi - p_lo) * d * d,
                 "%s: output row too long or row stride too small", who);
    const size_t lds = (size_t)(F * d + P) * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "%s: F=%d d=%d does not fit the LDS staging", who, F, d);
    if (d == 16 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && ld % 4 == 0)
        hipLaunchKernelGGL(pnn_outer_fwd16_kernel, dim3(B), dim3(256), lds, stream, emb, B, F, out, p_lo, p_hi, ld);
    else if (d % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && ld % 4 == 0)
        hipLaunchKernelGGL(pnn_outer_fwd_kernel<4>, dim3(B), dim3(256), lds, stream, emb, B, F, d, out, p_lo, p_hi, ld);
    else hipLaunchKernelGGL(pnn_outer_fwd_kernel<1>, dim3(B), dim3(256), lds, stream, emb, B, F, d, out, p_lo, p_hi, ld);
    RS_CHECK_LAUNCH(who);
    return RS_OK;
}

static int pnn_outer_bwd_impl(const float* emb, const float* dout, long long ld, int B, int F, int d, int i_lo, int i_hi,
                              int accumulate, float* demb, hipStream_t stream, const char* who) {
    RS_CHECK_ARG(B >= 0 && F > 1 && d > 0, "%s: bad sizes", who);
    RS_CHECK_ARG(0 <= i_lo && i_lo < i_hi && i_hi <= F, "%s: bad field range [%d, %d)", who, i_lo, i_hi);
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(emb && dout && demb, "%s: null argument", who);
    if (d == 16 && (size_t)6 * F * 16 * sizeof(float) <= LDS_LIMIT && (reinterpret_cast<uintptr_t>(dout) & 15) == 0 &&
        ld % 4 == 0) {
        hipLaunchKernelGGL(pnn_outer_bwd16_kernel, dim3(B), dim3(256), (size_t)6 * F * 16 * sizeof(float), stream, emb, dout, B,
                           F, demb, i_lo, i_hi, ld, accumulate);
        RS_CHECK_LAUNCH(who);
        return RS_OK;
    }
    const size_t lds = (size_t)F * d * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "%s: F*d=%d does not fit the LDS staging", who, F * d);
    hipLaunchKernelGGL(pnn_outer_bwd_kernel, dim3(B), dim3(256), lds, stream, emb, dout, B, F, d, demb, i_lo, i_hi, ld,
                       accumulate);
    RS_CHECK_LAUNCH(who);
    return RS_OK;
}

int rs_pnn_outer_fwd(const float* emb, int B, int F, int d, float* out, hipStream_t stream) {
    return pnn_outer_fwd_impl(emb, B, F, d, 0, F, out, (long long)(F * (F - 1) / 2) * d * d, stream, "rs_pnn_outer_fwd");
}

int rs_pnn_outer_bwd(const float* emb, const float* dout, int B, int F, int d, float* demb, hipStream_t stream) {
    return pnn_outer_bwd_impl(emb, dout, (long long)(F * (F - 1) / 2) * d * d, B, F, d, 0, F, 0, demb, stream,
                              "rs_pnn_outer_bwd");
}

int rs_pnn_outer_fwd_range(const float* emb, int B, int F, int d, int i_lo, int i_hi, float* out, long long ld,
                           hipStream_t stream) {
    return pnn_outer_fwd_impl(emb, B, F, d, i_lo, i_hi, out, ld, stream, "rs_pnn_outer_fwd_range");
}

int rs_pnn_outer_bwd_range(const float* emb, const float* dout, long long ld, int B, int F, int d, int i_lo, int i_hi,
                           int accumulate, float* demb, hipStream_t stream) {
    return pnn_outer_bwd_impl(emb, dout, ld, B, F, d, i_lo, i_hi, accumulate, demb, stream, "rs_pnn_outer_bwd_range");
}

// DCN cross network, L layers: x0 [B,D], w [L,D], bias [L,D] -> out [B,D]; S [B,L] (x_l . w_l) is saved
// for rs_dcn_cross_bwd (model/DCN/DCN.py:48-52).
int rs_dcn_cross_fwd(const float* x0, const float* w, const float* bias, int B, int D, int L, float* out, float* S,
                     hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && D > 0 && L > 0, "rs_dcn_cross_fwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(x0 && w && bias && out, "rs_dcn_cross_fwd: null argument");
    const int T = dcn_reg_T(D);
    if (T && L <= 4 && (size_t)2 * L * D * sizeof(float) <= LDS_LIMIT) {
        const int grid = rs_cdiv(B, 4) < 2048 ? rs_cdiv(B, 4) : 2048;
        const size_t rlds = (size_t)2 * L * D * sizeof(float);
#define RS_DCN_F(TT, LL) hipLaunchKernelGGL((dcn_cross_fwd_reg_kernel<TT, LL>), dim3(grid), dim3(256), rlds, stream, x0, w, bias, B, D, out, S)
        RS_DCN_DISPATCH(RS_DCN_F, T, L)
#undef RS_DCN_F
        RS_CHECK_LAUNCH("rs_dcn_cross_fwd");
        return RS_OK;
    }
    const size_t lds = (size_t)4 * D * sizeof(float);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_dcn_cross_fwd: D=%d does not fit the LDS staging", D);
    hipLaunchKernelGGL(dcn_cross_fwd_kernel, dim3(rs_cdiv(B, 4)), dim3(256), lds, stream, x0, w, bias, B, D, L, out, S);
    RS_CHECK_LAUNCH("rs_dcn_cross_fwd");
    return RS_OK;
}

// dx0 [B,D]; partials [rows, 2*L*D] with rows = rs_interact_partial_rows(B): per-row [dW | dB] partial sums,
// to be summed over rows by the caller.
static int dcn_cross_bwd_impl(const float* x0, const float* S, const float* dout, const float* w, const float* bias, int B,
                              int D, int L, float* dx0, float* partials, int rows, const float* dxadd, hipStream_t stream) {
    RS_CHECK_ARG(B > 0 && D > 0 && L > 0, "rs_dcn_cross_bwd: bad sizes");
    RS_CHECK_ARG(x0 && S && dout && w && bias && dx0 && partials, "rs_dcn_cross_bwd: null argument");
    RS_CHECK_ARG(rows == rs_interact_partial_rows(B), "rs_dcn_cross_bwd: partials must have %d rows",
                 rs_interact_partial_rows(B));
    const int T = dcn_reg_T(D);
    if (T && L <= 4 && (size_t)8 * L * D * sizeof(float) <= LDS_BIG) {
        const size_t rlds = (size_t)8 * L * D * sizeof(float);
#define RS_DCN_B(TT, LL)                                                                                               \
    do {                                                                                                               \
        RS_CHECK_ARG(raise_lds(reinterpret_cast<const void*>(dcn_cross_bwd_reg_kernel<TT, LL>)), "rs_dcn_cross_bwd: cannot raise the LDS limit"); \
        hipLaunchKernelGGL((dcn_cross_bwd_reg_kernel<TT, LL>), dim3(rows), dim3(256), rlds, stream, x0, S, dout, w, bias, B, D, \
                           dx0, partials, dxadd);                                                                      \
    } while (0)
        RS_DCN_DISPATCH(RS_DCN_B, T, L)
#undef RS_DCN_B
        RS_CHECK_LAUNCH("rs_dcn_cross_bwd");
        return RS_OK;
    }
    const size_t per_wave = (size_t)(2 * L + 2) * D * sizeof(float);
    RS_CHECK_ARG(per_wave <= LDS_BIG, "rs_dcn_cross_bwd: L=%d D=%d does not fit the LDS staging", L, D);
    const int nw = waves_for(per_wave);
    RS_CHECK_ARG(raise_lds(reinterpret_cast<const void*>(dcn_cross_bwd_kernel)), "rs_dcn_cross_bwd: cannot raise the LDS limit");
    hipLaunchKernelGGL(dcn_cross_bwd_kernel, dim3(rows), dim3(64 * nw), nw * per_wave, stream, x0, S, dout, w, bias, B, D, L,
                       dx0, partials, dxadd);
    RS_CHECK_LAUNCH("rs_dcn_cross_bwd");
    return RS_OK;
}

int rs_dcn_cross_bwd(const float* x0, const float* S, const float* dout, const float* w, const float* bias, int B, int D,
                     int L, float* dx0, float* partials, int rows, hipStream_t stream) {
    return dcn_cross_bwd_impl(x0, S, dout, w, bias, B, D, L, dx0, partials, rows, nullptr, stream);
}

// rs_dcn_cross_bwd with dx0 = (the cross network's gradient of x0) + dx_add [B, D]: x0 also feeds the deep tower
// (model/DCN/DCN.py:48-52), whose gradient of it arrives first -- the sum needs no launch of its own.  dx_add must not
// overlap dx0.
int rs_dcn_cross_bwd_add(const float* x0, const float* S, const float* dout, const float* w, const float* bias, int B, int D,
                         int L, const float* dx_add, float* dx0, float* partials, int rows, hipStream_t stream) {
    RS_CHECK_ARG(dx_add && dx_add != dx0, "rs_dcn_cross_bwd_add: null or aliased dx_add");
    return dcn_cross_bwd_impl(x0, S, dout, w, bias, B, D, L, dx0, partials, rows, dx_add, stream);
}

// NFM bi-interaction, reference form: E [B,D], V [D,K] -> out [B,K]; s_out [B,K] = E V is saved for the
// backward (model/NFM/NFM.py:42-44).
int rs_nfm_bi_fwd(const float* E, const float* V, int B, int D, int K, float* out, float* s_out, hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && D > 0 && K > 0 && K <= NFM_KMAX, "rs_nfm_bi_fwd: bad sizes (hidden_dim <= %d)", NFM_KMAX);
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(E && V && out, "rs_nfm_bi_fwd: null argument");
    {
        const int T = dcn_reg_T(D), KT = nfm_KT(K);
        const size_t rlds = (size_t)D * K * sizeof(float);
        if (T && rlds <= LDS_LIMIT) {
            const int grid = rs_cdiv(B, 4) < 2048 ? rs_cdiv(B, 4) : 2048;
#define RS_NFM_FR(TT, KK) hipLaunchKernelGGL((nfm_bi_fwd_reg_kernel<TT, KK>), dim3(grid), dim3(256), rlds, stream, E, V, B, D, K, out, s_out)
            RS_NFM_FWD_DISPATCH(RS_NFM_FR, T, KT)
#undef RS_NFM_FR
            RS_CHECK_LAUNCH("rs_nfm_bi_fwd");
            return RS_OK;
        }
    }
#define RS_NFM_F(KT) hipLaunchKernelGGL(nfm_bi_fwd_kernel<KT>, dim3(rs_cdiv(B, 4)), dim3(256), 0, stream, E, V, B, D, K, out, s_out)
    if (K <= 1) RS_NFM_F(1); else if (K <= 2) RS_NFM_F(2); else if (K <= 4) RS_NFM_F(4); else if (K <= 8) RS_NFM_F(8); else RS_NFM_F(16);
#undef RS_NFM_F
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
    {
        const int T = dcn_reg_T(D), KT = nfm_KT(K);
        const size_t rlds = (size_t)4 * D * K * sizeof(float);
        if (T && T * KT <= 64 && rlds <= LDS_BIG) {
#define RS_NFM_BR(TT, KK)                                                                                              \
    do {                                                                                                               \
        RS_CHECK_ARG(raise_lds(reinterpret_cast<const void*>(nfm_bi_bwd_reg_kernel<TT, KK>)), "rs_nfm_bi_bwd: cannot raise the LDS limit"); \
        hipLaunchKernelGGL((nfm_bi_bwd_reg_kernel<TT, KK>), dim3(rows), dim3(256), rlds, stream, E, V, s, dout, B, D, K, dE, \
                           partials);                                                                                  \
    } while (0)
            RS_NFM_BWD_DISPATCH(RS_NFM_BR, T, KT)
#undef RS_NFM_BR
            RS_CHECK_LAUNCH("rs_nfm_bi_bwd");
            return RS_OK;
        }
    }
    const size_t per_wave = (size_t)D * K * sizeof(float);
    RS_CHECK_ARG(per_wave <= LDS_BIG, "rs_nfm_bi_bwd: D*K=%d does not fit the LDS staging", D * K);
    const int nw = waves_for(per_wave);
#define RS_NFM_B(KT)                                                                                                   \
    do {                                                                                                               \
        RS_CHECK_ARG(raise_lds(reinterpret_cast<const void*>(nfm_bi_bwd_kernel<KT>)), "rs_nfm_bi_bwd: cannot raise the LDS limit"); \
        hipLaunchKernelGGL(nfm_bi_bwd_kernel<KT>, dim3(rows), dim3(64 * nw), nw * per_wave, stream, E, V, s, dout, B, D, K, dE, \
                           partials);                                                                                  \
    } while (0)
    if (K <= 1) RS_NFM_B(1); else if (K <= 2) RS_NFM_B(2); else if (K <= 4) RS_NFM_B(4); else if (K <= 8) RS_NFM_B(8); else RS_NFM_B(16);
#undef RS_NFM_B
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
    if (F <= 64 && afm_wave_shape(d, A) && (reinterpret_cast<uintptr_t>(emb) & 15) == 0) {
        const int region = d <= 16 ? afm_mx_region_floats(F, d, A, false) : afm_region_floats(F, d, A, false);
        const int wgrid = rs_cdiv(B, 4) < 2048 ? rs_cdiv(B, 4) : 2048;
#define RS_AFM_FW(KERN, DD, AA)                                                                                        \
    do {                                                                                                               \
        RS_CHECK_ARG(raise_lds(reinterpret_cast<const void*>(KERN<DD, AA>)), "rs_afm_fwd: cannot raise the LDS limit"); \
        hipLaunchKernelGGL((KERN<DD, AA>), dim3(wgrid), dim3(256), (size_t)4 * region * sizeof(float), stream, emb, B, \
                           F, W, bA, h, pw, pb, out, region);                                                          \
    } while (0)
#define RS_AFM_FW_M(DD, AA)                                                                                            \
    do {                                                                                                               \
        if (F <= 48) RS_AFM_FW(afm_fwd_mfma3, DD, AA);                                                                 \
        else RS_AFM_FW(afm_fwd_mfma4, DD, AA);                                                                         \
    } while (0)
#define RS_AFM_FW_V(DD, AA) RS_AFM_FW(afm_fwd_wave_kernel, DD, AA)
        RS_CHECK_ARG((size_t)4 * region * sizeof(float) <= LDS_BIG, "rs_afm_fwd: F=%d d=%d A=%d does not fit the LDS staging", F, d, A);
        RS_AFM_DISPATCH(RS_AFM_FW_M, RS_AFM_FW_V)
#undef RS_AFM_FW_M
#undef RS_AFM_FW_V
#undef RS_AFM_FW
        RS_CHECK_LAUNCH("rs_afm_fwd");
        return RS_OK;
    }
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
    if (F <= 64 && afm_wave_shape(d, A) && ((reinterpret_cast<uintptr_t>(emb) | reinterpret_cast<uintptr_t>(demb)) & 15) == 0) {
        const int region = d <= 16 ? afm_mx_region_floats(F, d, A, true) : afm_region_floats(F, d, A, true);
        RS_CHECK_ARG((size_t)4 * region * sizeof(float) <= LDS_BIG, "rs_afm_bwd: F=%d d=%d A=%d does not fit the LDS staging", F, d, A);
#define RS_AFM_BW(KERN, DD, AA)                                                                                        \
    do {                                                                                                               \
        RS_CHECK_ARG(raise_lds(reinterpret_cast<const void*>(KERN<DD, AA>)), "rs_afm_bwd: cannot raise the LDS limit"); \
        hipLaunchKernelGGL((KERN<DD, AA>), dim3(rows), dim3(256), (size_t)4 * region * sizeof(float), stream, emb,     \
                           dout, B, F, W, bA, h, pw, demb, partials, region);                                          \
    } while (0)
#define RS_AFM_BW_M(DD, AA)                                                                                            \
    do {                                                                                                               \
        if (F <= 48) RS_AFM_BW(afm_bwd_mfma3, DD, AA);                                                                 \
        else RS_AFM_BW(afm_bwd_mfma4, DD, AA);                                                                         \
    } while (0)
#define RS_AFM_BW_V(DD, AA) RS_AFM_BW(afm_bwd_wave_kernel, DD, AA)
        RS_AFM_DISPATCH(RS_AFM_BW_M, RS_AFM_BW_V)
#undef RS_AFM_BW_M
#undef RS_AFM_BW_V
#undef RS_AFM_BW
        RS_CHECK_LAUNCH("rs_afm_bwd");
        return RS_OK;
    }
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
    const int LP = d / 4;
    if (d % 4 == 0 && LP <= 16 && (LP & (LP - 1)) == 0 && F < 32768 && (reinterpret_cast<uintptr_t>(T) & 15) == 0) {
        const int P = F * (F - 1) / 2;
        const size_t vlds = (size_t)((P + 1) & ~1) * sizeof(int) + (size_t)4 * F * sizeof(long long);
        if (vlds <= LDS_LIMIT) {
            const int grid = rs_cdiv(B, 4) < 4096 ? rs_cdiv(B, 4) : 4096;
#define RS_FFM_F(L) hipLaunchKernelGGL(ffm_fwd_vec_kernel<L>, dim3(grid), dim3(256), vlds, stream, idx, row_offset, vocab, T, B, F, out, err_flag)
            switch (LP) {
                case 1: RS_FFM_F(1); break;
                case 2: RS_FFM_F(2); break;
                case 4: RS_FFM_F(4); break;
                case 8: RS_FFM_F(8); break;
                default: RS_FFM_F(16); break;
            }
#undef RS_FFM_F
            RS_CHECK_LAUNCH("rs_ffm_fwd");
            return RS_OK;
        }
    }
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

// FFM term and its gradient on gathered rows G [B, F, F*d] (see ffm_rows_fwd_kernel); d in {4, 8, 16, 32, 64}.
int rs_ffm_rows_fwd(const float* G, int B, int F, int d, float* out, hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 1 && F < 32768, "rs_ffm_rows_fwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(G && out && (reinterpret_cast<uintptr_t>(G) & 15) == 0, "rs_ffm_rows_fwd: null or unaligned argument");
    const int LP = d / 4;
    RS_CHECK_ARG(d % 4 == 0 && LP >= 1 && LP <= 16 && (LP & (LP - 1)) == 0, "rs_ffm_rows_fwd: embedding_dim %d not in {4,8,16,32,64}", d);
    const size_t lds = (size_t)(F * (F - 1) / 2) * sizeof(int);
    RS_CHECK_ARG(lds <= LDS_LIMIT, "rs_ffm_rows_fwd: F=%d does not fit the LDS pair table", F);
    const int grid = rs_cdiv(B, 4) < 4096 ? rs_cdiv(B, 4) : 4096;
#define RS_FFM_R(L) hipLaunchKernelGGL(ffm_rows_fwd_kernel<L>, dim3(grid), dim3(256), lds, stream, G, B, F, out)
    switch (LP) {
        case 1: RS_FFM_R(1); break;
        case 2: RS_FFM_R(2); break;
        case 4: RS_FFM_R(4); break;
        case 8: RS_FFM_R(8); break;
        default: RS_FFM_R(16); break;
    }
#undef RS_FFM_R
    RS_CHECK_LAUNCH("rs_ffm_rows_fwd");
    return RS_OK;
}

int rs_ffm_rows_bwd(const float* G, const float* dout, int B, int F, int d, float* dG, hipStream_t stream) {
    RS_CHECK_ARG(B >= 0 && F > 1 && d > 0 && d % 4 == 0, "rs_ffm_rows_bwd: bad sizes");
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(G && dout && dG && ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(dG)) & 15) == 0,
                 "rs_ffm_rows_bwd: null or unaligned argument");
    hipLaunchKernelGGL(ffm_rows_bwd_kernel, dim3(flat_blocks((long long)B * F * F * d / 4)), dim3(256), 0, stream, G, dout, B, F,
                       d, dG);
    RS_CHECK_LAUNCH("rs_ffm_rows_bwd");
    return RS_OK;
}

// Number of sum-of-squares partials rs_ffm_bwd_rows writes.
int rs_ffm_bwd_num_partials(int F, int cap) {
    return F * ((cap < FFM_ROW_GROUPS ? cap : FFM_ROW_GROUPS) + (cap < FFM_FINISH_GROUPS ? cap : FFM_FINISH_GROUPS));
}

// Compact gradients of the touched rows of the stacked FFM table: rowgrad [F, cap, F*d] (segment s of field f at
// [f, s, :], the layout rs_sparse_update / rs_sparse_scatter_dense read with row width F*d) and
// norm_partials [rs_ffm_bwd_num_partials(F, cap)].  chunk_part [F, cap, F*d] is scratch.  plan: rs_sparse_bwd_sort of
// the same columnar idx with capacity cap and window L <= 64.  d % 4 == 0 and F*d <= 1024.
int rs_ffm_bwd_rows(const int* plan, const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, const float* T,
                    const float* dout, int B, int F, int cap, int d, float* rowgrad, float* chunk_part,
                    float* norm_partials, hipStream_t stream) {
    RS_CHECK_ARG(B > 0 && F > 1 && d > 0 && cap >= B, "rs_ffm_bwd_rows: bad sizes");
    RS_CHECK_ARG(d % 4 == 0 && F * d <= 1024, "rs_ffm_bwd_rows: needs embedding_dim %% 4 == 0 and F*d <= 1024 (F=%d d=%d)", F, d);
    RS_CHECK_ARG(plan && idx && row_offset && vocab && T && dout && rowgrad && chunk_part && norm_partials,
                 "rs_ffm_bwd_rows: null argument");
    RS_CHECK_ARG(((reinterpret_cast<uintptr_t>(T) | reinterpret_cast<uintptr_t>(rowgrad) |
                   reinterpret_cast<uintptr_t>(chunk_part)) & 15) == 0,
                 "rs_ffm_bwd_rows: table, rowgrad and chunk_part must be 16-byte aligned");
    const PlanView pv = carve_plan(const_cast<int*>(plan), F, cap);
    const int g1 = cap < FFM_ROW_GROUPS ? cap : FFM_ROW_GROUPS, g2 = cap < FFM_FINISH_GROUPS ? cap : FFM_FINISH_GROUPS;
    hipLaunchKernelGGL(ffm_bwd_chunks_kernel, dim3(g1, F), dim3(256), 0, stream, pv, idx, row_offset, vocab, T, dout, B, F,
                       cap, d, rowgrad, chunk_part, norm_partials);
    hipLaunchKernelGGL(ffm_bwd_finish_kernel, dim3(g2, F), dim3(256), 0, stream, pv, F, cap, d, rowgrad, chunk_part,
                       norm_partials + (size_t)F * g1);
    RS_CHECK_LAUNCH("rs_ffm_bwd_rows");
    return RS_OK;
}

}  // extern "C"
