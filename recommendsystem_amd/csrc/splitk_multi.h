// Reducing launch for several split-K products whose slabs wait in their workspaces, shared by the stand-alone entry
// point (gemm.hip) and by the launch that runs it beside the first phase of the embedding gradient reduction
// (sparse.hip).
#pragma once
#include "common.h"

namespace {

// The same for several deferred products in one launch (no epilogue): product p owns workgroups
// [first_block[p], first_block[p+1]); float4 where the buffers allow it.
constexpr int SK_MAX = 8;
struct SplitKMultiArgs {
    const float* ws[SK_MAX]; float* C[SK_MAX];
    long long total[SK_MAX];
    int splitk[SK_MAX], vec[SK_MAX], first_block[SK_MAX + 1];
    int count;
};
// `block`: index of this workgroup among the a.first_block[a.count] reducing workgroups.
__device__ __forceinline__ void splitk_reduce_multi_body(const SplitKMultiArgs& a, int block) {
    int p = 0;
    while (p + 1 < a.count && block >= a.first_block[p + 1]) ++p;
    const int nb = a.first_block[p + 1] - a.first_block[p], b = block - a.first_block[p];
    const long long total = a.total[p], stride = (long long)nb * 256;
    const float* ws = a.ws[p];
    float* C = a.C[p];
    const int sk = a.splitk[p];
    if (a.vec[p]) {
        const long long t4 = total / 4;
        for (long long e = (long long)b * 256 + threadIdx.x; e < t4; e += stride) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int z = 0; z < sk; ++z) {
                const float4 t = reinterpret_cast<const float4*>(ws + z * total)[e];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            reinterpret_cast<float4*>(C)[e] = v;
        }
    } else {
        for (long long e = (long long)b * 256 + threadIdx.x; e < total; e += stride) {
            float v = 0.f;
            for (int z = 0; z < sk; ++z) v += ws[z * total + e];
            C[e] = v;
        }
    }
}

// Fills the argument block from host arrays; returns the number of reducing workgroups or -1 on a bad product.
inline int splitk_multi_fill(SplitKMultiArgs& a, const float* const* ws, float* const* C, const long long* total,
                             const int* splitk, int count) {
    int blocks = 0;
    for (int p = 0; p < count; ++p) {
        if (!(ws[p] && C[p] && total[p] > 0 && splitk[p] > 0)) return -1;
        a.ws[p] = ws[p]; a.C[p] = C[p]; a.total[p] = total[p]; a.splitk[p] = splitk[p];
        const bool al = ((reinterpret_cast<uintptr_t>(ws[p]) | reinterpret_cast<uintptr_t>(C[p])) & 15) == 0;
        a.vec[p] = (al && total[p] % 4 == 0) ? 1 : 0;
        a.first_block[p] = blocks;
        const long long units = a.vec[p] ? total[p] / 4 : total[p];
        const long long b = (units + 255) / 256;
        blocks += (int)(b > 1024 ? 1024 : b);
    }
    a.first_block[count] = blocks;
    a.count = count;
    return blocks;
}

}  // namespace
