// Device side of K1 (see gather.hip for the layout notes), shared by the stand-alone gather launch and by the
// launch that runs the gather beside the per-field index sort (sparse.hip).
#pragma once
#include "common.h"

namespace {

struct GatherArgs {
    const int64_t* idx;         // [F, B]
    const int64_t* row_offset;  // [F]
    const int64_t* vocab;       // [F]
    const float* table;         // [R, d] or null
    const float* table1;        // [R]    or null (1-dim "LR" tables)
    const float* bias;          // [1]    or null
    float* emb_out;             // [B, F, d] or null
    float* sum_out;             // [B, d]    or null : S = sum_f e_f
    float* bi_out;              // [B, d]    or null : 0.5*(S^2 - sum_f e_f^2)
    float* lr_out;              // [B]       or null : sum_f w_f[idx] + bias
    float* fm_out;              // [B]       or null : 0.5*sum_k(S_k^2 - sum_f e_fk^2)
    int* err_flag;              // or null; set to 1 when an index is outside [0, vocab)
    int B, F, d, G, C;
    // SLOTS form (the requester side of the row-sharded exchange): the row of entry (f, b) is slots[f*B + b] of a
    // buffer of received rows instead of row_offset[f] + idx[f, b] of a table; `table` then points at that buffer with
    // ld floats between rows and `table1` at its 1-dim column with ld1 floats between rows.
    const int* slots;
    int ld, ld1;
    // device scalar every looked-up value is multiplied by (null: none): tables kept as w / scale by a trainer that
    // applies a per-step shrink of ALL rows (L2 regularisation under SGD) to the scalar instead of the rows
    const float* scale;
};

template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<1> { using type = float; };

template <int VEC>
__device__ __forceinline__ void vload(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        *p = v[0];
    }
}

constexpr int U = 8;  // fields in flight per lane

// `wave`: index of this wave among all gather waves of the launch.
template <int VEC, bool SLOTS = false>
__device__ __forceinline__ void gather_fused_body(const GatherArgs& a, int wave) {
    const int lane = threadIdx.x & 63;
    const int lps = a.G * a.C;            // lanes per sample
    const int spw = 64 / lps;             // samples per wave
    const int sl = lane / lps;
    const int r = lane - sl * lps;
    const int g = r / a.C;
    const int c = r - g * a.C;
    const int lane_base = sl * lps;
    const long long b = (long long)wave * spw + sl;
    const bool active = (sl < spw) && (b < a.B);
    const int d = a.d;

    float acc[VEC], accq[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[e] = 0.f; accq[e] = 0.f; }
    float lr = 0.f;
    bool bad = false;
    const float sc = a.scale ? *a.scale : 1.f;

    const int iters = (a.F + a.G - 1) / a.G;
    for (int i0 = 0; i0 < iters; i0 += U) {
        long long row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = (i0 + u) * a.G + g;
            row[u] = -1;
            if (active && (i0 + u) < iters && f < a.F) {
                if constexpr (SLOTS) {
                    row[u] = a.slots[(long long)f * a.B + b];
                } else {
                    long long ix = a.idx[(long long)f * a.B + b];
                    if (ix < 0 || ix >= a.vocab[f]) { bad = true; ix = 0; }
                    row[u] = a.row_offset[f] + ix;
                }
            }
        }
        float v[U][VEC];
        float w1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[u][e] = 0.f;
            w1[u] = 0.f;
            if (row[u] >= 0) {
                if constexpr (SLOTS) {
                    if (a.table) vload<VEC>(a.table + row[u] * a.ld + c * VEC, v[u]);
                    if (a.table1 && c == 0) w1[u] = a.table1[row[u] * a.ld1];
                } else {
                    if (a.table) vload<VEC>(a.table + row[u] * d + c * VEC, v[u]);
                    if (a.table1 && c == 0) w1[u] = a.table1[row[u]];
                }
            }
        }
        if (a.scale) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[u][e] *= sc;
                w1[u] *= sc;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (row[u] >= 0) {
                const int f = (i0 + u) * a.G + g;
                if (a.emb_out) vstore<VEC>(a.emb_out + ((long long)b * a.F + f) * d + c * VEC, v[u]);
#pragma unroll
                for (int e = 0; e < VEC; ++e) { acc[e] += v[u][e]; accq[e] = fmaf(v[u][e], v[u][e], accq[e]); }
                lr += w1[u];
            }
        }
    }
    if (bad && a.err_flag) atomicOr(a.err_flag, 1);

    const bool need_red = a.sum_out || a.bi_out || a.lr_out || a.fm_out;
    if (!need_red) return;   // uniform across the grid

    // combine the G field groups (G is a power of two); every lane takes part in the shuffles
    for (int m = 1; m < a.G; m <<= 1) {
        const int src = lane_base + ((g ^ m) * a.C + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            acc[e] += __shfl(acc[e], src, 64);
            accq[e] += __shfl(accq[e], src, 64);
        }
        lr += __shfl(lr, src, 64);
    }
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) t += acc[e] * acc[e] - accq[e];
    // sum t over the C chunks of the sample (C need not be a power of two)
    float tsum = 0.f;
    for (int cc = 0; cc < a.C; ++cc) tsum += __shfl(t, lane_base + cc, 64);

    if (active && g == 0) {
        if (a.sum_out) vstore<VEC>(a.sum_out + b * d + c * VEC, acc);
        if (a.bi_out) {
            float bi[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) bi[e] = 0.5f * (acc[e] * acc[e] - accq[e]);
            vstore<VEC>(a.bi_out + b * d + c * VEC, bi);
        }
        if (c == 0) {
            if (a.lr_out) a.lr_out[b] = lr + (a.bias ? a.bias[0] : 0.f);
            if (a.fm_out) a.fm_out[b] = 0.5f * tsum;
        }
    }
}

int pick_layout(int d, int* VEC, int* G, int* C) {
    if (d % 4 == 0 && d / 4 <= 64) { *VEC = 4; *C = d / 4; }
    else if (d <= 64) { *VEC = 1; *C = d; }
    else return -1;
    int g = 8;
    while (g > 1 && g * (*C) > 32) g >>= 1;
    if (g * (*C) > 64) return -1;
    *G = g;
    return 0;
}

}  // namespace
