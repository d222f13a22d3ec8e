// Finishing launch for partial column sums left by several producers (GEMM epilogues, the 1-wide head), shared by
// the stand-alone entry point (gemm.hip) and by the launch that runs it beside the first phase of the embedding
// gradient reduction (sparse.hip).
#pragma once
#include "common.h"

namespace {

// Several column-sum reductions in one launch: problem p adds the `rows[p]` partial rows part[p][r*ld[p] + n]
// (n < n[p]) in a fixed order into out[p][n].  16 columns per workgroup: thread (c = tid % 16, r = tid / 16) adds
// rows r, r+16, ... (sixteen accumulators, combined pairwise), then the 16 row groups of a column meet in LDS.
constexpr int CF_MAX = 8;
struct ColsumFinishArgs {
    const float* part[CF_MAX]; float* out[CF_MAX];
    int rows[CF_MAX], n[CF_MAX], ld[CF_MAX], first_block[CF_MAX + 1];
    int count;
};
// `block`: index of this workgroup among the a.first_block[a.count] finishing workgroups; red: 16*17 floats of LDS.
__device__ __forceinline__ void colsum_finish_body(const ColsumFinishArgs& a, int block, float (*red)[17]) {
    int p = 0;
    while (p + 1 < a.count && block >= a.first_block[p + 1]) ++p;
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int n = (block - a.first_block[p]) * 16 + c;
    const float* part = a.part[p];
    const int rows = a.rows[p], ld = a.ld[p];
    // 16 independent loads in flight per thread: with a thousand partial rows (a 1-wide head at batch 16384) and only
    // n/16 workgroups, the loop is a chain of memory latencies, not a stream
    constexpr int CF_U = 16;
    float s[CF_U];
#pragma unroll
    for (int u = 0; u < CF_U; ++u) s[u] = 0.f;
    if (n < a.n[p]) {
        for (int g = r; g < rows; g += 16 * CF_U) {
#pragma unroll
            for (int u = 0; u < CF_U; ++u)
                if (g + 16 * u < rows) s[u] += part[(long long)(g + 16 * u) * ld + n];
        }
    }
#pragma unroll
    for (int w = CF_U / 2; w >= 1; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u) s[u] += s[u + w];
    red[r][c] = s[0];
    __syncthreads();
    if (r == 0 && n < a.n[p]) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][c];
        a.out[p][n] = t;
    }
}

// Fills the argument block from host arrays; returns the number of finishing workgroups or -1 on a bad problem.
inline int colsum_finish_fill(ColsumFinishArgs& a, const float* const* part, const int* rows, const int* n, const int* ld,
                              float* const* out, int count) {
    int blocks = 0;
    for (int p = 0; p < count; ++p) {
        if (!(part[p] && out[p] && rows[p] >= 0 && n[p] > 0 && ld[p] >= n[p])) return -1;
        a.part[p] = part[p]; a.out[p] = out[p]; a.rows[p] = rows[p]; a.n[p] = n[p]; a.ld[p] = ld[p];
        a.first_block[p] = blocks;
        blocks += rs_cdiv(n[p], 16);
    }
    a.first_block[count] = blocks;
    a.count = count;
    return blocks;
}

}  // namespace
