// K2e: fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bit-for-bit an fmaf chain in k order), with the MLP epilogues fused.
//
//   C[m, n] = epi( sum_k A(m, k) * B(k, n) )
//
// Operand storage (both fp32, row strides in elements):
//   A_MC = 0 : A(m, k) = A[m*lda + k]   (k contiguous; activations [M, K])
//   A_MC = 1 : A(m, k) = A[k*lda + m]   (m contiguous; used for dW = dZ^T X)
//   B_NC = 0 : B(k, n) = B[n*ldb + k]   (k contiguous; nn.Linear weight [out, in])
//   B_NC = 1 : B(k, n) = B[k*ldb + n]   (n contiguous; used for dX = dZ W and dW = dZ^T X)
//
// The three products of an MLP layer (model/utils/Layers.py:42 forward; its autograd backward):
//   forward  H  = act(X W^T + b)            A_MC=0 B_NC=0, epilogue bias + activation
//   dX/dZ'   dZ'= (dZ W) * act'(H_prev)     A_MC=0 B_NC=1, epilogue multiplies by act' of the
//                                           previous layer's output
//   dW       dW = dZ^T X                    A_MC=1 B_NC=1, split over the batch (K) dimension,
//                                           partial slabs summed in fixed order (deterministic)
//
// Structure: 256 threads = 4 waves in a 2x2 grid, block tile BM x BN (64 or 128 each), BK = 32.
// Both operand tiles are staged global -> registers -> LDS in k-major form S[k][m] so that an
// MFMA operand is one conflict-free ds_read_b32 per lane (lanes run along m, the lane half picks
// k); the next tile's global loads are issued before the MFMAs of the current tile and written to
// the other LDS buffer after them (one barrier per k-tile).
#include "common.h"
#include "philox.h"
#include "colsum_finish.h"
#include "splitk_multi.h"
#include "plan.h"
#include "sort_body.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias;      // [N] or null
    const float* mulsrc;    // [M, ldm] or null: C *= act_grad(mul_act, h = mulsrc[m, n])
    int M, N, K, lda, ldb, ldc, ldm;
    int act, mul_act;
    int kchunk;             // k range per blockIdx.z (multiple of BK)
    int splitk;
    float* ws;              // [splitk, M, N] partial slabs when splitk > 1
    float* cs_part;         // [ceil(M / (BM/2)), N] or null: column sums of the finished C per wave row
                            // (partial bias gradients when C is a dZ), added up by rs_colsum_finish_multi
    int vec_epi;            // C (and mulsrc, bias) allow float4 access: the LDS-staged epilogue is used
    unsigned* trace;        // experiment builds: per-wave cycle counters (see RS_GEMM_DBG bit 64)
    int dbg;                // ablation experiments (builds with -DRS_GEMM_EXPERIMENT read RS_GEMM_DBG):
                            // 1 = no DMA in the k loop, 2 = no stores, 4 = no MFMA, 8 = no barrier in the k loop,
                            // 16 = no LDS fragment reads after the first k-tile
    int defer;              // splitk > 1: leave the slabs in ws, the caller reduces them later (rs_splitk_reduce_multi)
    int accum;              // add what C holds to the product before bias and activation (a sum built up over several calls)
    int has_drop;           // multiply the activated result by the dropout mask of `drop` (element index m*N + n), before
    DropKey drop;           // the mulsrc factor: forward of a block with dropout, or the dX product that feeds such a block
};
#ifdef RS_GEMM_EXPERIMENT
#define RS_DBG(g) ((g).dbg)
#else
#define RS_DBG(g) 0
#endif

__device__ __forceinline__ float act_grad_from_h(int act, float h) {
    switch (act) {
        case RS_ACT_RELU: return h > 0.f ? 1.f : 0.f;
        case RS_ACT_TANH: return 1.f - h * h;
        case RS_ACT_LEAKYRELU: return h > 0.f ? 1.f : 0.01f;
        default: return 1.f;
    }
}

__device__ __forceinline__ float epilogue(const GemmArgs& g, float v, int m, int n) {
    if (g.accum) v += g.C[(long long)m * g.ldc + n];
    if (g.bias) v += g.bias[n];
    v = rs_act(g.act, v, 0.f);
    if (g.has_drop) {
        const long long flat = (long long)m * g.N + n;
        float f[4];
        drop_factors(drop_consts(g.drop), flat >> 2, f);
        const float fe = f[flat & 3];
        v = fe != 0.f ? v * fe : 0.f;
    }
    if (g.mulsrc) v *= act_grad_from_h(g.mul_act, g.mulsrc[(long long)m * g.ldm + n]);
    return v;
}


// Output-tile coordinates of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (each with its
// own L2), so the linear block id is first remapped so that the blocks sharing an XCD take one contiguous chunk
// of the tile sequence (bijective for any grid size), and the sequence itself walks the grid in groups of 8
// tile rows: the blocks resident on an XCD at one time then share A row panels and B column panels in its L2.
// Where a workgroup stands in its product: grid of gx x gy output tiles, lin = its linear id in that grid (x fastest),
// z = its slice of a split reduction.
struct Blk { int gx, gy, lin, z; };
__device__ __forceinline__ Blk blk_of_launch() {
    return Blk{(int)gridDim.x, (int)gridDim.y, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)blockIdx.z};
}

// Position of linear block id `orig` (of nwg ids dealt round-robin over the 8 XCDs) in a sequence in which every XCD
// owns one contiguous chunk (bijective for any nwg).
__device__ __forceinline__ int xcd_chunked(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
// Tile t of the walk over a gx x gy tile grid in groups of 8 tile rows, column by column inside a group.
__device__ __forceinline__ void grouped_walk(int t, int gx, int gy, int& tm, int& tn) {
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * gx;
    const int group = t / per_group, first_m = group * GROUP_M;
    const int gsz = min(gy - first_m, GROUP_M);
    const int rem = t - group * per_group;
    tm = first_m + rem % gsz;
    tn = rem / gsz;
}
// bk.lin < 0: bk.gx already holds the tile column and bk.gy the tile row (the caller did the mapping).
__device__ __forceinline__ void tile_coords(const Blk& bk, int& tm, int& tn) {
    if (bk.lin < 0) { tn = bk.gx; tm = bk.gy; return; }
    grouped_walk(xcd_chunked(bk.lin, bk.gx * bk.gy), bk.gx, bk.gy, tm, tn);
}

// ---- tile epilogue shared by both kernels ----------------------------------------------------------
// C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// splitk == 1: epilogue + store; optionally the column sums of the stored values over the wave's BM/2 rows.
// splitk  > 1: raw slab store (splitk_reduce_kernel adds the slabs and applies the epilogue).
template <int BM, int BN>
__device__ __forceinline__ void finish_tile(const GemmArgs& g, f32x16 (&acc)[BM / 64][BN / 64], int m0, int n0, int wm0,
                                            int wn0, int li, int lh, int wid, float* smem) {
    constexpr int TM = BM / 64, TN = BN / 64;
    const bool partial = g.splitk > 1;
    float* out = partial ? g.ws + (long long)blockIdx.z * g.M * g.N : g.C;
    const int ldo = partial ? g.N : g.ldc;
    float cs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) cs[j] = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < g.M && n < g.N) {
                    float v = acc[i][j][r];
                    if (!partial) v = epilogue(g, v, m, n);
                    out[(long long)m * ldo + n] = v;
                    cs[j] += v;
                }
            }
        }
    if (partial || !g.cs_part) return;
    // column sums of the workgroup's BM rows: lane halves, then the two wave rows through LDS
#pragma unroll
    for (int j = 0; j < TN; ++j) cs[j] += __shfl_xor(cs[j], 32, 64);
    __syncthreads();                                     // the staging LDS is free
    if (lh == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) smem[(wid >> 1) * BN + wn0 + j * 32 + li] = cs[j];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < BN && n0 + t < g.N) g.cs_part[(long long)(m0 / BM) * g.N + n0 + t] = smem[t] + smem[BN + t];
}

// ---- tile staging -----------------------------------------------------------------------------
// R = rows of the tile along the non-k dimension (BM or BN).  MC: that dimension is contiguous
// in global memory.  V: 16-byte loads are legal (alignment checked on the host).
template <int R, bool MC, bool V> struct Stage {
    static constexpr int LD = MC ? R + 4 : R + 1;
    static constexpr int NV = V ? R / 32 : R / 8;       // registers (float4 or float) per thread
    using reg_t = typename std::conditional<V, float4, float>::type;

    __device__ static __forceinline__ void load(reg_t (&r)[NV], const float* __restrict__ P, int ld, int rows,
                                                int row0, int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if constexpr (V && !MC) {          // 8 lanes cover 32 consecutive k of one row
                const int row = row0 + (tid >> 3) + i * 32, k = k0 + (tid & 7) * 4;
                r[i] = (row < rows && k < kend) ? *reinterpret_cast<const float4*>(P + (long long)row * ld + k)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
            } else if constexpr (V && MC) {    // consecutive lanes along the contiguous dimension
                const int e = tid + i * 256, k = k0 + e / (R / 4), row = row0 + (e % (R / 4)) * 4;
                r[i] = (row < rows && k < kend) ? *reinterpret_cast<const float4*>(P + (long long)k * ld + row)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
            } else if constexpr (!V && !MC) {
                const int e = tid + i * 256, k = k0 + (e & 31), row = row0 + (e >> 5);
                r[i] = (row < rows && k < kend) ? P[(long long)row * ld + k] : 0.f;
            } else {
                const int e = tid + i * 256, k = k0 + e / R, row = row0 + (e % R);
                r[i] = (row < rows && k < kend) ? P[(long long)k * ld + row] : 0.f;
            }
        }
    }
    __device__ static __forceinline__ void store(const reg_t (&r)[NV], float* S, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if constexpr (V && !MC) {
                const int row = (tid >> 3) + i * 32, k = (tid & 7) * 4;
                S[(k + 0) * LD + row] = r[i].x;
                S[(k + 1) * LD + row] = r[i].y;
                S[(k + 2) * LD + row] = r[i].z;
                S[(k + 3) * LD + row] = r[i].w;
            } else if constexpr (V && MC) {
                const int e = tid + i * 256, k = e / (R / 4), row = (e % (R / 4)) * 4;
                *reinterpret_cast<float4*>(S + k * LD + row) = r[i];
            } else if constexpr (!V && !MC) {
                const int e = tid + i * 256;
                S[(e & 31) * LD + (e >> 5)] = r[i];
            } else {
                const int e = tid + i * 256;
                S[(e / R) * LD + (e % R)] = r[i];
            }
        }
    }
};

template <int BM, int BN, bool A_MC, bool B_NC, bool VA, bool VB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    using SA = Stage<BM, A_MC, VA>;
    using SB = Stage<BN, B_NC, VB>;
    constexpr int A_STAGE = BK * SA::LD, B_STAGE = BK * SB::LD;
    constexpr int TM = BM / 64, TN = BN / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * A_STAGE;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wid >> 1) * (BM / 2), wn0 = (wid & 1) * (BN / 2);
    int tile_m, tile_n;
    tile_coords(blk_of_launch(), tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    typename SA::reg_t ra[SA::NV];
    typename SB::reg_t rb[SB::NV];
    if (nk > 0) {
        SA::load(ra, g.A, g.lda, g.M, m0, kbeg, kend, tid);
        SB::load(rb, g.B, g.ldb, g.N, n0, kbeg, kend, tid);
        SA::store(ra, As, tid);
        SB::store(rb, Bs, tid);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            SA::load(ra, g.A, g.lda, g.M, m0, kbeg + (kt + 1) * BK, kend, tid);
            SB::load(rb, g.B, g.ldb, g.N, n0, kbeg + (kt + 1) * BK, kend, tid);
        }
        const float* as = As + cur * A_STAGE + wm0 + li;
        const float* bs = Bs + cur * B_STAGE + wn0 + li;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[(kk * 2 + lh) * SA::LD + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[(kk * 2 + lh) * SB::LD + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            SA::store(ra, As + (cur ^ 1) * A_STAGE, tid);
            SB::store(rb, Bs + (cur ^ 1) * B_STAGE, tid);
        }
        __syncthreads();
    }

    finish_tile<BM, BN>(g, acc, m0, n0, wm0, wn0, li, lh, wid, smem);
}

// =================================================================================================
// Fast path: LDS ring filled by LDS-DMA (global_load_lds_dwordx4), NST stages, NST-1 k-tiles in flight
// across the barrier, one barrier per k-tile.  No staging registers, no ds_write.
//
// Matrix instruction: v_mfma_f32_16x16x4_f32.  Measured on MI355X with a register-only loop
// (tools/mfma_clock.hip): it sustains its nominal 32 cycles per SIMD (137 TFLOP/s chip-wide at the
// 2.17 GHz the part holds under this load), while v_mfma_f32_32x32x2_f32 needs about 80 cycles for its
// nominal 64 (about 100 TFLOP/s), whatever the number of independent accumulator chains or waves.
// Operand roles: lane (r = lane & 15, g = lane >> 4) supplies A[row r][k_g] and B[k_g][col r]; the result
// block has col = lane & 15, row = 4*(lane >> 4) + reg.
//
// LDS images (per stage, per operand, BK = 32):
//   k-contiguous operand  ("RK"): [rows][32 k] row-major, 128 B per row, 16-byte chunks XOR-swizzled
//                                 (chunk c of row r is stored at chunk c ^ ((r>>1)&7)); an MFMA operand
//                                 read is one conflict-free ds_read_b128 = 4 k-steps of one row.
//   m/n-contiguous operand ("KM"): [32 k][rows], row positions XOR-swizzled by the k row
//                                 (element (k, r) sits at position r ^ (((k>>2)&3) << 4)), so that the four
//                                 lane groups of a ds_read_b32 -- four k rows 1 KiB apart -- hit different banks.
// The DMA writes LDS lane-linearly (wave base + lane*16 B), so both swizzles are applied to each lane's
// SOURCE address and again on the read.  Lanes whose source would fall outside the matrix (row / k
// tails) read a 16-byte block of zeros instead.
// k assignment inside a tile: half h = 0..1, step j = 0..3, lane group g  ->  k = 16h + 4g + j, the same
// for both operands.
__device__ __attribute__((aligned(16))) float g_zero_src[4] = {0.f, 0.f, 0.f, 0.f};

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RS_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define RS_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int R, bool MC> struct DmaOp {
    static constexpr int P = R / 32;                 // DMA instructions per thread per tile
    static constexpr int BYTES = R * 32 * 4;
    static constexpr int NREAD = MC ? 2 : 1;         // ds_read instructions per 16-row block per half k-tile

    // A lane's source address for k-tile t is  base + t*step + voff[p]  with base and step uniform over the
    // workgroup (scalar registers) and voff a fixed per-lane byte offset, so issuing a DMA costs no vector
    // arithmetic.  Rows past the end of the matrix are clamped to its last rows: what they load only feeds
    // result rows/columns that are never stored.
    const char* base;
    long long step;       // bytes per k-tile
    unsigned voff[P];
    int kofs[P];          // this lane's k offset inside a tile (for the K tail)

    __device__ __forceinline__ void init(const float* __restrict__ Pm, int ld, int rows, int row0, int kbeg, int tid) {
        if constexpr (!MC) {
            base = reinterpret_cast<const char*>(Pm + (long long)row0 * ld + kbeg);
            step = 32 * 4;
        } else {
            base = reinterpret_cast<const char*>(Pm + (long long)kbeg * ld + row0);
            step = (long long)32 * ld * 4;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int pos = p * 256 + tid;
            if constexpr (!MC) {
                const int row = pos >> 3, cp = pos & 7, c = cp ^ ((row >> 1) & 7);
                kofs[p] = c * 4;
                const int rc = min(row, rows - 1 - row0);
                voff[p] = (unsigned)(rc * ld + c * 4) * 4u;
            } else {
                const int k = pos / (R / 4), r4 = ((pos % (R / 4)) * 4) ^ (((k >> 2) & 3) << 4);
                kofs[p] = k;
                const int rc = min(r4, rows - 4 - row0);
                voff[p] = (unsigned)(k * ld + rc) * 4u;
            }
        }
    }
    // Issue DMA number p of k-tile t into the operand image at LDS byte address `dst` (uniform).
    // FULL: the tile lies inside [kbeg, kend): scalar base + per-lane offset, no predicate.
    // Otherwise lanes whose k is past kend read zeros (first k of the tile = k0).
    template <int p, bool FULL>
    __device__ __forceinline__ void issue_one(int t, int k0, int kend, unsigned dst, int wid) {
        const char* cur = base + (long long)t * step;
        const unsigned d = dst + (unsigned)((p * 256 + wid * 64) * 16);
        if constexpr (FULL) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(d), "v"(voff[p]), "s"(cur) : "memory");
        } else {
            const char* s = (k0 + kofs[p] < kend) ? cur + voff[p] : reinterpret_cast<const char*>(g_zero_src);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "s"(d), "v"(s) : "memory");
        }
    }
};

// Operand fragments of one half k-tile (16 k): 4 floats (steps j = 0..3) per 16-row block.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool MC, int T> struct Frag {
    f32x4 v[T];          // RK: one ds_read_b128
    f32x2 s[T][2];       // KM: two ds_read2st64_b32 (steps 0,1 | 2,3)
    __device__ __forceinline__ float get(int t, int j) const {
        if constexpr (MC) return s[t][j >> 1][j & 1];
        else return v[t][j];
    }
};

// KM read of k rows (16H + 2J) and (16H + 2J + 1) above the lane's group row: row stride R floats = R/64
// units of 64 dwords
template <int R, int H, int J>
__device__ __forceinline__ void km_read2(f32x2& d, unsigned a) {
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3"
                 : "=v"(d) : "v"(a), "i"((16 * H + 2 * J) * (R / 64)), "i"((16 * H + 2 * J + 1) * (R / 64)));
}

template <int R, bool MC, int T, int H>
__device__ __forceinline__ void frag_read(Frag<MC, T>& f, const unsigned (&addr)[T][2], unsigned so) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if constexpr (!MC) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(f.v[t]) : "v"(addr[t][H] + so));
        } else {
            const unsigned a = addr[t][0] + so;          // includes the 4g k rows of this lane group
            km_read2<R, H, 0>(f.s[t][0], a);
            km_read2<R, H, 1>(f.s[t][1], a);
        }
    }
}

// Tie every destination register of the asm reads to a point after the wait (the compiler must not
// touch them earlier); emits no instruction.
template <bool MC, int T>
__device__ __forceinline__ void frag_pin(Frag<MC, T>& f) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if constexpr (!MC) asm volatile("" : "+v"(f.v[t]));
        else asm volatile("" : "+v"(f.s[t][0]), "+v"(f.s[t][1]));
    }
}

// ---- epilogues for the 16x16 result blocks (same contracts as finish_tile / finish_tile_vec) ----------
template <int BM, int BN>
__device__ __forceinline__ void finish16(const GemmArgs& g, f32x4 (&acc)[BM / 32][BN / 32], int m0, int n0, int wm0, int wn0,
                                         int lr, int lg, int wid, float* smem, int z) {
    constexpr int RB = BM / 32, CB = BN / 32;
    const bool partial = g.splitk > 1;
    float* out = partial ? g.ws + (long long)z * g.M * g.N : g.C;
    const int ldo = partial ? g.N : g.ldc;
    float cs[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) cs[j] = 0.f;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int n = n0 + wn0 + j * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm0 + i * 16 + 4 * lg + r;
                if (m < g.M && n < g.N) {
                    float v = acc[i][j][r];
                    if (!partial) v = epilogue(g, v, m, n);
                    out[(long long)m * ldo + n] = v;
                    cs[j] += v;
                }
            }
        }
    if (partial || !g.cs_part) return;
    // column sums of the workgroup's BM rows: the four lane groups, then the two wave rows through LDS
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        cs[j] += __shfl_xor(cs[j], 16, 64);
        cs[j] += __shfl_xor(cs[j], 32, 64);
    }
    if (lg == 0) {
#pragma unroll
        for (int j = 0; j < CB; ++j) smem[(wid >> 1) * BN + wn0 + j * 16 + lr] = cs[j];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < BN && n0 + t < g.N) g.cs_part[(long long)(m0 / BM) * g.N + n0 + t] = smem[t] + smem[BN + t];
}

// 16-byte aligned outputs: the workgroup's tile goes through LDS so that every lane stores (and, for the act'
// factor, loads) float4s and a wave covers whole 256/512-byte row segments.
// smem: BM*(BN+4) + 4*BN floats, no longer read by any wave (the caller synchronises).
template <int BM, int BN>
__device__ __forceinline__ void finish16_vec(const GemmArgs& g, f32x4 (&acc)[BM / 32][BN / 32], int m0, int n0, int wm0,
                                             int wn0, int lr, int lg, float* smem, int z) {
    constexpr int RB = BM / 32, CB = BN / 32;
    constexpr int LDT = BN + 4;                      // staged row stride (keeps float4 alignment, spreads banks)
    constexpr int LPR = BN / 4;                      // lanes per staged row
    constexpr int RPP = 256 / LPR;                   // rows per pass of the 256 threads
    const bool partial = g.splitk > 1;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) smem[(wm0 + i * 16 + 4 * lg + r) * LDT + wn0 + j * 16 + lr] = acc[i][j][r];
    __syncthreads();
    const int tid = threadIdx.x;
    const int c4 = (tid % LPR) * 4, rr = tid / LPR;
    const int n = n0 + c4;
    float* out = partial ? g.ws + (long long)z * g.M * g.N : g.C;
    const int ldo = partial ? g.N : g.ldc;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!partial && g.bias && n < g.N) bv = *reinterpret_cast<const float4*>(g.bias + n);
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    DropConsts dc{};
    if (!partial && g.has_drop) dc = drop_consts(g.drop);
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
        const int row = p * RPP + rr, m = m0 + row;
        if (m < g.M && n < g.N) {
            float4 v = *reinterpret_cast<const float4*>(smem + row * LDT + c4);
            if (!partial) {
                if (g.accum) {
                    const float4 o = *reinterpret_cast<const float4*>(out + (long long)m * ldo + n);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                v.x = rs_act(g.act, v.x + bv.x, 0.f); v.y = rs_act(g.act, v.y + bv.y, 0.f);
                v.z = rs_act(g.act, v.z + bv.z, 0.f); v.w = rs_act(g.act, v.w + bv.w, 0.f);
                if (g.has_drop) {                      // N % 4 == 0 here: the float4 is one block of the mask
                    float f[4];
                    drop_factors(dc, ((long long)m * g.N + n) >> 2, f);
                    v.x = f[0] != 0.f ? v.x * f[0] : 0.f; v.y = f[1] != 0.f ? v.y * f[1] : 0.f;
                    v.z = f[2] != 0.f ? v.z * f[2] : 0.f; v.w = f[3] != 0.f ? v.w * f[3] : 0.f;
                }
                if (g.mulsrc) {
                    const float4 h = *reinterpret_cast<const float4*>(g.mulsrc + (long long)m * g.ldm + n);
                    v.x *= act_grad_from_h(g.mul_act, h.x); v.y *= act_grad_from_h(g.mul_act, h.y);
                    v.z *= act_grad_from_h(g.mul_act, h.z); v.w *= act_grad_from_h(g.mul_act, h.w);
                }
            }
            *reinterpret_cast<float4*>(out + (long long)m * ldo + n) = v;
            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
        }
    }
    if (partial || !g.cs_part) return;
    // column sums: rows of one wave (lanes LPR apart), then the four waves through LDS
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
        cs.x += __shfl_xor(cs.x, o, 64); cs.y += __shfl_xor(cs.y, o, 64);
        cs.z += __shfl_xor(cs.z, o, 64); cs.w += __shfl_xor(cs.w, o, 64);
    }
    float* red = smem + BM * LDT;                    // [4][BN]
    if ((tid & 63) < LPR) *reinterpret_cast<float4*>(red + (tid >> 6) * BN + c4) = cs;
    __syncthreads();
    if (tid < BN && n0 + tid < g.N)
        g.cs_part[(long long)(m0 / BM) * g.N + n0 + tid] = (red[tid] + red[BN + tid]) + (red[2 * BN + tid] + red[3 * BN + tid]);
}

template <int BM, int BN, bool A_MC, bool B_NC, int NST>
__device__ __forceinline__ void gemm_dma_body(const GemmArgs& g, const Blk bk, float* smem) {
    using DA = DmaOp<BM, A_MC>;
    using DB = DmaOp<BN, B_NC>;
    constexpr int DIST = NST - 1;                  // k-tiles in flight ahead of the one being multiplied
    constexpr int ST_BYTES = DA::BYTES + DB::BYTES;
    constexpr int RB = BM / 32, CB = BN / 32;      // 16x16 result blocks per wave: (BM/2) x (BN/2) wave tile
    constexpr int NDMA = DA::P + DB::P;
    constexpr int NRD_ = RB * DA::NREAD + CB * DB::NREAD;
    constexpr int NRD = NRD_ > 15 ? 15 : NRD_;     // LDS reads complete in order: waiting for fewer is still right

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int wm0 = (wid >> 1) * (BM / 2), wn0 = (wid & 1) * (BN / 2);
    int tile_m, tile_n;
    tile_coords(bk, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = bk.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + 31) / 32;

    DA da; DB db;
    da.init(g.A, g.lda, g.M, m0, kbeg, tid);
    db.init(g.B, g.ldb, g.N, n0, kbeg, tid);

    // LDS read addresses (bytes from the start of a stage), per 16-row block and half k-tile
    const unsigned lds0 = (unsigned)(unsigned long long)RS_LDS_PTR(smem);
    unsigned aA[RB][2], aB[CB][2];
#pragma unroll
    for (int t = 0; t < RB; ++t) {
        const int r = wm0 + t * 16 + lr;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            aA[t][h] = lds0 + (A_MC ? (unsigned)(((4 * lg) * BM + (r ^ (lg << 4))) * 4)
                                    : (unsigned)((r * 8 + ((4 * h + lg) ^ ((r >> 1) & 7))) * 16));
    }
#pragma unroll
    for (int t = 0; t < CB; ++t) {
        const int r = wn0 + t * 16 + lr;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            aB[t][h] = lds0 + DA::BYTES + (B_NC ? (unsigned)(((4 * lg) * BN + (r ^ (lg << 4))) * 4)
                                                : (unsigned)((r * 8 + ((4 * h + lg) ^ ((r >> 1) & 7))) * 16));
    }

    f32x4 acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // DMA number `i` (0 .. NDMA-1) of k-tile kt into stage st; A's pieces first, then B's.
    auto issue_i = [&](auto ic, auto fullc, int kt, int st) {
        constexpr int i = decltype(ic)::value;
        constexpr bool FULL = decltype(fullc)::value;
        if ((RS_DBG(g) & 1) && kt >= DIST) return;
        const unsigned base = lds0 + (unsigned)(st * ST_BYTES);
        if constexpr (i < DA::P) da.template issue_one<i, FULL>(kt, kbeg + kt * 32, kend, base, wid);
        else db.template issue_one<i - DA::P, FULL>(kt, kbeg + kt * 32, kend, base + DA::BYTES, wid);
    };
    auto issue_all = [&](int kt, int st) {
        constexpr std::false_type F{};
        issue_i(std::integral_constant<int, 0>{}, F, kt, st);
        issue_i(std::integral_constant<int, 1>{}, F, kt, st);
        issue_i(std::integral_constant<int, 2>{}, F, kt, st);
        issue_i(std::integral_constant<int, 3>{}, F, kt, st);
        if constexpr (NDMA > 4) { issue_i(std::integral_constant<int, 4>{}, F, kt, st); issue_i(std::integral_constant<int, 5>{}, F, kt, st); }
        if constexpr (NDMA > 6) { issue_i(std::integral_constant<int, 6>{}, F, kt, st); issue_i(std::integral_constant<int, 7>{}, F, kt, st); }
    };
    // prologue: DIST tiles in flight (tiles past the end read zeros, which keeps the counts uniform)
    issue_all(0, 0);
    if constexpr (DIST > 1) issue_all(1, 1);
    if constexpr (DIST > 2) issue_all(2, 2);

#ifdef RS_GEMM_EXPERIMENT
#define RS_CYC() ((unsigned)__builtin_amdgcn_s_memtime() & 0xFFFFF)      /* shader clock; only used where no LDS read is pending */
    unsigned tr_start = RS_CYC(), tr_dma = 0, tr_bar = 0, tr_lds = 0, tr_first = 0;
#define RS_TR(acc, t0) { const unsigned t1__ = RS_CYC(); acc += (t1__ - (t0)) & 0xFFFFF; t0 = t1__; }
#else
#define RS_TR(acc, t0)
#endif
    // One k-tile.  FULL: the tile issued during it (kt + DIST) lies entirely inside the k range.
    auto body = [&](auto fullc, int kt, int st) {
        constexpr std::integral_constant<bool, decltype(fullc)::value> FC{};
        // tile kt has landed once only the younger DMAs (tiles kt+1 .. kt+DIST-1) are outstanding
#ifdef RS_GEMM_EXPERIMENT
        unsigned tc = RS_CYC();
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (DIST - 1)) : "memory");
        RS_TR(tr_dma, tc);
        if (!(RS_DBG(g) & 8)) __builtin_amdgcn_s_barrier();
        RS_TR(tr_bar, tc);
#ifdef RS_GEMM_EXPERIMENT
        if (kt == 0) tr_first = (tc - tr_start) & 0xFFFFF;
#endif
        const unsigned so = (unsigned)(st * ST_BYTES);
        const int nst = (st + DIST) % NST;           // stage read in iteration kt-1: free after the barrier
        Frag<A_MC, RB> fa[2];
        Frag<B_NC, CB> fb[2];
        // One half k-tile: the other half's LDS reads, wait for this half's, then 4 steps of RB*CB MFMAs with
        // the DMAs of k-tile kt+DIST spread between the steps so that their issue overlaps MFMA time.
        auto half = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            if constexpr (h == 0) {
                if (!((RS_DBG(g) & 16) && kt > 0)) {
                    frag_read<BM, A_MC, RB, 1>(fa[1], aA, so);
                    frag_read<BN, B_NC, CB, 1>(fb[1], aB, so);
                }
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            frag_pin<A_MC, RB>(fa[h]);
            frag_pin<B_NC, CB>(fb[h]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!(RS_DBG(g) & 4)) {
#pragma unroll
                    for (int tm = 0; tm < RB; ++tm)
#pragma unroll
                        for (int tn = 0; tn < CB; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[h].get(tm, j), fb[h].get(tn, j), acc[tm][tn],
                                                                               0, 0, 0);
                }
                // 8 issue slots per k-tile (one after every step); NDMA <= 8 of them used
                if (4 * h + j < NDMA) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (j == 0) issue_i(std::integral_constant<int, (4 * h + 0 < NDMA ? 4 * h + 0 : 0)>{}, FC, kt + DIST, nst);
                    if (j == 1) issue_i(std::integral_constant<int, (4 * h + 1 < NDMA ? 4 * h + 1 : 0)>{}, FC, kt + DIST, nst);
                    if (j == 2) issue_i(std::integral_constant<int, (4 * h + 2 < NDMA ? 4 * h + 2 : 0)>{}, FC, kt + DIST, nst);
                    if (j == 3) issue_i(std::integral_constant<int, (4 * h + 3 < NDMA ? 4 * h + 3 : 0)>{}, FC, kt + DIST, nst);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (!((RS_DBG(g) & 16) && kt > 0)) {
            frag_read<BM, A_MC, RB, 0>(fa[0], aA, so);
            frag_read<BN, B_NC, CB, 0>(fb[0], aB, so);
        }
        half(std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{});
    };
    // Steady state: groups of NST k-tiles whose look-ahead tiles are all full (no predicates, compile-time
    // stages); then the remaining tiles one by one, predicated only when their look-ahead tile is not full.
    const int nfull = (kend - kbeg) / 32;
    int kt = 0;
    for (; kt + NST - 1 + DIST < nfull; kt += NST) {
        body(std::true_type{}, kt, 0);
        body(std::true_type{}, kt + 1, 1);
        if constexpr (NST > 2) body(std::true_type{}, kt + 2, 2);
        if constexpr (NST > 3) body(std::true_type{}, kt + 3, 3);
    }
    for (; kt < nk; ++kt) {
        if (kt + DIST < nfull) body(std::true_type{}, kt, kt % NST);
        else body(std::false_type{}, kt, kt % NST);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the trailing (zero) tiles
#ifdef RS_GEMM_EXPERIMENT
    const unsigned tr_loop_end = RS_CYC();
    if ((RS_DBG(g) & 64) && g.trace) {
        __syncthreads();
        if (RS_DBG(g) & 128) {} else {
            if (g.vec_epi) finish16_vec<BM, BN>(g, acc, m0, n0, wm0, wn0, lr, lg, smem, bk.z);
            else finish16<BM, BN>(g, acc, m0, n0, wm0, wn0, lr, lg, wid, smem, bk.z);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tr_end = RS_CYC();
        if (lane == 0) {
            unsigned* o = g.trace + (bk.lin * 4 + wid) * 8;
            o[0] = tr_start; o[1] = tr_first; o[2] = (tr_loop_end - tr_start) & 0xFFFFF; o[3] = (tr_end - tr_start) & 0xFFFFF;
            o[4] = tr_dma; o[5] = tr_bar; o[6] = tr_lds; o[7] = nk;
        }
        return;
    }
#endif

    if (RS_DBG(g) & 2) { if (acc[0][0][0] == 12345.678f) g.C[0] = 1.f; return; }
    __syncthreads();                                      // every wave is done reading the LDS ring
    if (g.vec_epi) finish16_vec<BM, BN>(g, acc, m0, n0, wm0, wn0, lr, lg, smem, bk.z);
    else finish16<BM, BN>(g, acc, m0, n0, wm0, wn0, lr, lg, wid, smem, bk.z);
}

template <int BM, int BN, bool A_MC, bool B_NC, int NST>
__global__ __launch_bounds__(256) void gemm_f32_dma_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_dma_body<BM, BN, A_MC, B_NC, NST>(g, blk_of_launch(), smem);
}

// Several independent products in ONE launch (64x64 tiles, NST-stage ring): the workgroups of product p take the linear
// block ids [first[p], first[p] + tiles[p] * splitk[p]), slice-major; the starts are multiples of 8, so a product's
// tiles keep the XCD a stand-alone launch would give them.  Products that wait for the same predecessor and whose
// results nobody reads before the next launch (a layer's dX and dW, which both need only its dZ) fill the chip
// together instead of one after the other with a ramp and a tail each.
constexpr int GROUP_MAX = 4;
struct GemmGroupArgs {
    GemmArgs g[GROUP_MAX];
    int first[GROUP_MAX + 1];
    int gx[GROUP_MAX], gy[GROUP_MAX];
    int layout[GROUP_MAX];      // 0: A, B k-contiguous; 1: A k-contiguous, B n-contiguous; 2: A m-contiguous, B n-contiguous
    int count;
};

// workgroup b of a grouped launch
template <int NST>
__device__ __forceinline__ void group_block(const GemmGroupArgs& a, int b, float* smem) {
    int p = 0;
#pragma unroll
    for (int q = 1; q < GROUP_MAX; ++q)
        if (q < a.count && b >= a.first[q]) p = q;
    const int local = b - a.first[p];
    const int tiles = a.gx[p] * a.gy[p], nwg = tiles * a.g[p].splitk;
    if (local >= nwg) return;                           // padding up to the next product's start
    // An XCD takes one contiguous chunk of the (slice, tile) sequence: with the reduction split, the workgroups that
    // share an L2 then work inside one or two slices and read each operand panel of their slice once, instead of every
    // slice being spread over all eight L2s.
    const int t = xcd_chunked(local, nwg);
    const int z = t / tiles;
    int tm, tn;
    grouped_walk(t - z * tiles, a.gx[p], a.gy[p], tm, tn);
    const Blk bk{tn, tm, -1, z};
    const int lay = a.layout[p];
    if (lay == 0) gemm_dma_body<64, 64, false, false, NST>(a.g[p], bk, smem);
    else if (lay == 1) gemm_dma_body<64, 64, false, true, NST>(a.g[p], bk, smem);
    else gemm_dma_body<64, 64, true, true, NST>(a.g[p], bk, smem);
}

template <int NST>
__global__ __launch_bounds__(256) void gemm_f32_group_kernel(GemmGroupArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    group_block<NST>(a, blockIdx.x, smem);
}

// A grouped launch whose first `lead` workgroups (a multiple of 8, so that the products keep their XCDs) sort the index
// lists of the NEXT batch, one field each (sort_body.h, 256 threads): the sort needs nothing but that batch, takes F
// workgroups for some tens of microseconds and fits the LDS a GEMM workgroup has anyway -- beside a block's backward
// products it costs the step nothing, where a launch of its own is that long on 39 of the 256 compute units.
// The sort workgroup shares its compute unit with two GEMM workgroups whose waves mostly wait for the matrix pipeline:
// at equal priority they still take two thirds of the issue slots and the sort -- one wave per SIMD, bound by
// instruction issue -- takes three times as long as alone, longer than the products it rides with.  With priority its
// short vector / scalar / LDS instructions go first and cost the matrix pipeline nothing it would have used.
template <int ITEMS>
__device__ __forceinline__ void sort_rider(const SortArgs* sa, int f, unsigned long long* buf) {
    __builtin_amdgcn_s_setprio(3);
#ifdef RS_SORT_TIMING
    const unsigned long long t0 = wall_clock64();
#endif
    field_sort_body<ITEMS, 256>(*sa, f, buf);
#ifdef RS_SORT_TIMING
    if (threadIdx.x == 0) {           // behind the spilled items of all fields (experiment builds only)
        unsigned long long* dbg = sa->spill + (size_t)sa->F * sa->cap + (sa->phase == 2 ? 2 * sa->F : 0);
        dbg[2 * f] = t0; dbg[2 * f + 1] = wall_clock64();
    }
#endif
}

template <int NST, int ITEMS>
__global__ __launch_bounds__(256) void gemm_f32_group_sort_kernel(GemmGroupArgs a, SortArgs sa, int lead) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < lead) {
        if (b < sa.F) sort_rider<ITEMS>(&sa, b, reinterpret_cast<unsigned long long*>(smem));
        return;
    }
    group_block<NST>(a, b - lead, smem);
}

// Sum the split-K slabs in slab order and apply the epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g) {
    const long long total = (long long)g.M * g.N;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        float v = 0.f;
        for (int z = 0; z < g.splitk; ++z) v += g.ws[z * total + e];
        const int m = (int)(e / g.N), n = (int)(e % g.N);
        g.C[(long long)m * g.ldc + n] = epilogue(g, v, m, n);
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(SplitKMultiArgs a) {
    splitk_reduce_multi_body(a, blockIdx.x);
}

template <int BM, int BN, bool A_MC, bool B_NC, bool VA, bool VB>
int launch(const GemmArgs& g, hipStream_t stream) {
    using SA = Stage<BM, A_MC, VA>;
    using SB = Stage<BN, B_NC, VB>;
    constexpr size_t smem = (size_t)2 * BK * (SA::LD + SB::LD) * sizeof(float);
    auto kern = gemm_f32_kernel<BM, BN, A_MC, B_NC, VA, VB>;
    static bool attr_set = false;
    if (!attr_set && smem > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess) {
            rs_set_error("rs_gemm_f32: cannot raise dynamic LDS limit to %zu", smem);
            return RS_ERR_LAUNCH;
        }
        attr_set = true;
    }
    dim3 grid(rs_cdiv(g.N, BN), rs_cdiv(g.M, BM), g.splitk), block(256);
    hipLaunchKernelGGL(kern, grid, block, smem, stream, g);
    RS_CHECK_LAUNCH("rs_gemm_f32");
    if (g.splitk > 1 && !g.defer) {
        const long long total = (long long)g.M * g.N;
        int blocks = (int)min((long long)2048, (total + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g);
        RS_CHECK_LAUNCH("rs_gemm_f32(split-k reduce)");
    }
    return RS_OK;
}

template <int BM, int BN, bool A_MC, bool B_NC, int NST = 3>
int launch_dma(const GemmArgs& g, hipStream_t stream) {
    constexpr size_t smem = (size_t)NST * (BM + BN) * 32 * sizeof(float);
    auto kern = gemm_f32_dma_kernel<BM, BN, A_MC, B_NC, NST>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess) {
            rs_set_error("rs_gemm_f32: cannot raise dynamic LDS limit to %zu", smem);
            return RS_ERR_LAUNCH;
        }
        attr_set = true;
    }
    dim3 grid(rs_cdiv(g.N, BN), rs_cdiv(g.M, BM), g.splitk), block(256);
    hipLaunchKernelGGL(kern, grid, block, smem, stream, g);
    RS_CHECK_LAUNCH("rs_gemm_f32");
    if (g.splitk > 1 && !g.defer) {
        const long long total = (long long)g.M * g.N;
        int blocks = (int)min((long long)2048, (total + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g);
        RS_CHECK_LAUNCH("rs_gemm_f32(split-k reduce)");
    }
    return RS_OK;
}

template <bool A_MC, bool B_NC>
int dispatch(const GemmArgs& g, bool va, bool vb, int tile, hipStream_t s) {
    if (va && vb) {
        switch (tile) {
            case 0: return launch_dma<128, 128, A_MC, B_NC>(g, s);
            case 1: return launch_dma<128, 64, A_MC, B_NC>(g, s);
            case 2: return launch_dma<64, 128, A_MC, B_NC>(g, s);
            case 3: return launch_dma<64, 64, A_MC, B_NC>(g, s);
            case 5: return launch_dma<64, 64, A_MC, B_NC, 2>(g, s);        // 64x64 with a 2-stage ring (5 workgroups/CU)
            default: return launch<64, 64, A_MC, B_NC, true, true>(g, s);   // tile 4: register-staged kernel
        }
    }
    if (va) return launch<64, 64, A_MC, B_NC, true, false>(g, s);
    if (vb) return launch<64, 64, A_MC, B_NC, false, true>(g, s);
    return launch<64, 64, A_MC, B_NC, false, false>(g, s);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// column sums: out[n] = sum_m scale[m] * X[m, n] (scale optional).  Stage 1: one workgroup per
// (256 columns, 64 rows): every lane keeps 16 independent 16-byte loads in flight; the four waves'
// sums meet in LDS.  Stage 2 adds the per-row-block partials in order.  Fixed summation order.
constexpr int CS_ROWS = 64;
template <bool VEC>
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const float* __restrict__ X, int M, int N, int ld,
                                                            const float* __restrict__ scale,
                                                            float* __restrict__ part_out) {
    __shared__ float part[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int mbeg = blockIdx.y * CS_ROWS, mend = min(M, mbeg + CS_ROWS);
    const int n0 = blockIdx.x * 256;
    if constexpr (VEC) {
        const int n = n0 + lane * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) {
            float4 v[16];
            float sc[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int m = mbeg + w + u * 4;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                sc[u] = 1.f;
                if (m < mend) {
                    v[u] = *reinterpret_cast<const float4*>(X + (long long)m * ld + n);
                    if (scale) sc[u] = scale[m];
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc.x = fmaf(sc[u], v[u].x, acc.x); acc.y = fmaf(sc[u], v[u].y, acc.y);
                acc.z = fmaf(sc[u], v[u].z, acc.z); acc.w = fmaf(sc[u], v[u].w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(&part[w][lane * 4]) = acc;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = n0 + c * 64 + lane;
            float acc = 0.f;
            if (n < N)
                for (int m = mbeg + w; m < mend; m += 4) acc = fmaf(scale ? scale[m] : 1.f, X[(long long)m * ld + n], acc);
            part[w][c * 64 + lane] = acc;
        }
    }
    __syncthreads();
    const int n = n0 + threadIdx.x;
    if (n < N) {
        const int t = threadIdx.x;
        part_out[(long long)blockIdx.y * N + n] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    }
}
// Stage 2: one workgroup per 64 columns; wave r adds partial rows r, r+4, ... (all loads of a lane are
// independent), then the four waves' sums meet in LDS.
__global__ __launch_bounds__(256) void colsum_stage2_kernel(const float* __restrict__ part, int chunks, int N,
                                                            float* __restrict__ out) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        for (int k = r; k < chunks; k += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k + 4 * u < chunks) s[u] += part[(long long)(k + 4 * u) * N + n];
        }
    }
    red[r][c] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (r == 0 && n < N) out[n] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

__global__ __launch_bounds__(256) void colsum_finish_multi_kernel(ColsumFinishArgs a) {
    __shared__ float red[16][17];
    colsum_finish_body(a, blockIdx.x, red);
}

}  // namespace

// One product of rs_gemm_f32_group (field for field the struct of the same name in include/rs_hip.h).
extern "C" {
typedef struct rs_gemm_desc {
    int a_mc, b_nc, M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias; int act;
    const float* mulsrc; int ldm; int mul_act;
    int splitk;
    float* workspace; size_t workspace_bytes;
    float* colsum_part;
    /* drop_p > 0 (splitk == 1 only): the mask of rs_gemm_f32_drop with this key is multiplied into the result */
    float drop_p; unsigned drop_salt; unsigned long long drop_seed; unsigned long long drop_call_host;
    const long long* drop_call_dev;
} rs_gemm_desc;
}

extern "C" {

size_t rs_gemm_workspace_bytes(int M, int N, int splitk) {
    return splitk > 1 ? (size_t)splitk * M * N * sizeof(float) : 0;
}

// Tile/split choice for a problem on a device with `cus` compute units.
// Returns the tile code (0: 128x128, 1: 128x64, 2: 64x128, 3: 64x64, all with a 3-stage ring; 4: the
// register-staged 64x64 kernel; 5: 64x64 with a 2-stage ring) and writes the split count.
int rs_gemm_plan(int M, int N, int K, int cus, int* splitk_out) {
    // Measured on MI355X at the MLP shapes of this workload (M or K = 4096; N, K in 256..1024), tools/gemm_tiles.py:
    // unsplit products run best on 64x64 tiles with the 2-stage ring (32 KiB of LDS: five workgroups per CU, so
    // the 1024-tile layers finish in one round); products whose tile grid leaves CUs idle split the reduction
    // into a power-of-two number of slices so that about three workgroups per CU exist, each slice keeping at
    // least 512 of k, and use the deeper 3-stage ring.
    const long long blocks = (long long)rs_cdiv(M, 64) * rs_cdiv(N, 64);
    int sk = 1;
    if (blocks < 2LL * cus && K >= 1024) {
        const long long want = (3LL * cus + blocks - 1) / blocks;
        while (sk * 2 <= want && sk * 2 <= K / 512 && sk * 2 <= 16) sk *= 2;
    }
    *splitk_out = sk;
    return sk > 1 ? 3 : 5;
}

// Rows of the partial column sums a GEMM with tile code `tile` leaves for an M-row output (one per tile row).
int rs_gemm_colsum_rows(int M, int tile) { return rs_cdiv(M, (tile == 0 || tile == 1) ? 128 : 64); }

// C = epi(op(A) op(B)); see the header of this file for operand conventions.
// tile < 0 and splitk <= 0 let the library choose.  workspace is needed when the resolved split count is > 1
// (rs_gemm_workspace_bytes with the same splitk, or sized for splitk = 16).
// colsum_part (optional, [rs_gemm_colsum_rows(M, tile), N], needs an explicit tile and a resolved splitk of 1):
// receives per-wave-row column sums of the finished C; their sum over rows (rs_colsum_finish_multi) is
// sum_m C[m, n] -- the bias gradient when C is a dZ.
static int gemm_impl(int a_mc, int b_nc, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int act, const float* mulsrc, int ldm, int mul_act,
                     int tile, int splitk, float* workspace, size_t workspace_bytes, float* colsum_part, int defer,
                     int accum, const DropKey* drop, hipStream_t stream) {
    RS_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "rs_gemm_f32: negative size");
    if (M == 0 || N == 0) return RS_OK;
    RS_CHECK_ARG(A && B && C, "rs_gemm_f32: null operand");
    RS_CHECK_ARG(!(a_mc && !b_nc), "rs_gemm_f32: layout (A m-contiguous, B k-contiguous) is not provided");
    RS_CHECK_ARG(lda >= (a_mc ? M : K) && ldb >= (b_nc ? N : K) && ldc >= N, "rs_gemm_f32: bad leading dims");
    RS_CHECK_ARG(!mulsrc || ldm >= N, "rs_gemm_f32: bad ldm");
    RS_CHECK_ARG(act >= RS_ACT_NONE && act <= RS_ACT_LEAKYRELU, "rs_gemm_f32: activation %d not fusable", act);
    RS_CHECK_ARG(!colsum_part || tile >= 0, "rs_gemm_f32: colsum_part needs an explicit tile (its row count depends on it)");
    int cus = 256;
    if (tile < 0 || splitk <= 0) {
        int sk = 1;
        int t = rs_gemm_plan(M, N, K, cus, &sk);
        if (tile < 0) tile = t;
        if (splitk <= 0) splitk = sk;
    }
    if (splitk > 1) {
        int maxsk = rs_cdiv(K, BK);
        if (splitk > maxsk) splitk = maxsk > 0 ? maxsk : 1;
    }
    if (splitk > 1)
        RS_CHECK_ARG(workspace && workspace_bytes >= rs_gemm_workspace_bytes(M, N, splitk),
                     "rs_gemm_f32: workspace too small for splitk=%d (%zu bytes given)", splitk, workspace_bytes);
    RS_CHECK_ARG(!colsum_part || splitk == 1, "rs_gemm_f32: colsum_part needs an unsplit reduction (splitk resolved to %d)",
                 splitk);
    static const int dbg = getenv("RS_GEMM_DBG") ? atoi(getenv("RS_GEMM_DBG")) : 0;
    static const bool allow_vec = !(getenv("RS_GEMM_VEC_EPI") && atoi(getenv("RS_GEMM_VEC_EPI")) == 0);
    const bool vec_epi = allow_vec && N % 4 == 0 && aligned16(C) && ldc % 4 == 0 && (!bias || aligned16(bias)) &&
                         (!mulsrc || (aligned16(mulsrc) && ldm % 4 == 0)) && (splitk == 1 || aligned16(workspace));
    GemmArgs g{A, B, C, bias, mulsrc, M, N, K, lda, ldb, ldc, ldm, act, mul_act, 0, splitk, workspace, colsum_part,
               vec_epi ? 1 : 0, (dbg & 64) ? reinterpret_cast<unsigned*>(workspace) : nullptr, dbg, defer, accum};
    if (drop && drop->p > 0.f) {
        RS_CHECK_ARG(drop->p <= 1.f, "rs_gemm_f32_drop: dropout probability has to be between 0 and 1, but got %f", (double)drop->p);
        g.has_drop = 1;
        g.drop = *drop;
    }
    RS_CHECK_ARG(!defer || splitk > 1, "rs_gemm_f32_slabs: the reduction did not split (splitk resolved to %d)", splitk);
    g.kchunk = rs_cdiv(rs_cdiv(K, splitk), BK) * BK;
    if (g.kchunk == 0) g.kchunk = BK;
    const bool va = aligned16(A) && lda % 4 == 0 && (a_mc ? M % 4 == 0 : K % 4 == 0);
    const bool vb = aligned16(B) && ldb % 4 == 0 && (b_nc ? N % 4 == 0 : K % 4 == 0);
    RS_CHECK_ARG(!colsum_part || (tile != 0 && tile != 1) || (va && vb),
                 "rs_gemm_f32: colsum_part with 128-row tiles needs 16-byte aligned operands");
    if (!a_mc && !b_nc) return dispatch<false, false>(g, va, vb, tile, stream);
    if (!a_mc && b_nc) return dispatch<false, true>(g, va, vb, tile, stream);
    return dispatch<true, true>(g, va, vb, tile, stream);
}

int rs_gemm_f32(int a_mc, int b_nc, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                float* C, int ldc, const float* bias, int act, const float* mulsrc, int ldm, int mul_act,
                int tile, int splitk, float* workspace, size_t workspace_bytes, float* colsum_part,
                hipStream_t stream) {
    return gemm_impl(a_mc, b_nc, M, N, K, A, lda, B, ldb, C, ldc, bias, act, mulsrc, ldm, mul_act, tile, splitk, workspace,
                     workspace_bytes, colsum_part, 0, 0, nullptr, stream);
}

// rs_gemm_f32 with the dropout mask of rs_dropout_apply(p, seed, salt, call_host, call_dev) multiplied into the result
// (element index m*N + n) after bias and activation and before the mulsrc factor: the forward product of a block whose
// output goes through dropout, and the dX product whose result is the gradient of such an output.
int rs_gemm_f32_drop(int a_mc, int b_nc, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int act, const float* mulsrc, int ldm, int mul_act,
                     int tile, int splitk, float* workspace, size_t workspace_bytes, float* colsum_part, float drop_p,
                     unsigned long long drop_seed, unsigned drop_salt, unsigned long long drop_call_host,
                     const long long* drop_call_dev, hipStream_t stream) {
    const DropKey key{drop_p, drop_salt, drop_seed, drop_call_host, drop_call_dev};
    return gemm_impl(a_mc, b_nc, M, N, K, A, lda, B, ldb, C, ldc, bias, act, mulsrc, ldm, mul_act, tile, splitk, workspace,
                     workspace_bytes, colsum_part, 0, 0, &key, stream);
}

// C = act(C + op(A) op(B) + bias): one term of a product whose reduction dimension is walked in several calls (the
// operands of a later term may then reuse the buffers of an earlier one); bias and activation belong to the last call.
int rs_gemm_f32_acc(int a_mc, int b_nc, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                    int ldc, const float* bias, int act, int tile, int splitk, float* workspace, size_t workspace_bytes,
                    hipStream_t stream) {
    return gemm_impl(a_mc, b_nc, M, N, K, A, lda, B, ldb, C, ldc, bias, act, nullptr, 0, 0, tile, splitk, workspace,
                     workspace_bytes, nullptr, 0, 1, nullptr, stream);
}

// The split-K product without its reducing launch: the splitk slabs [splitk, M, N] stay in `workspace` (which must
// stay untouched until they are reduced) and C is written by a later rs_splitk_reduce_multi, which serves several
// such products in one launch.  tile >= 0 and splitk > 1 are required; no epilogue.
int rs_gemm_f32_slabs(int a_mc, int b_nc, int M, int N, int K, const float* A, int lda, const float* B, int ldb, int tile,
                      int splitk, float* workspace, size_t workspace_bytes, hipStream_t stream) {
    RS_CHECK_ARG(tile >= 0 && splitk > 1 && workspace, "rs_gemm_f32_slabs: needs an explicit tile, splitk > 1 and a workspace");
    RS_CHECK_ARG(splitk <= rs_cdiv(K, BK), "rs_gemm_f32_slabs: splitk %d exceeds the k-tiles of K=%d", splitk, K);
    return gemm_impl(a_mc, b_nc, M, N, K, A, lda, B, ldb, workspace, N, nullptr, RS_ACT_NONE, nullptr, 0, 0, tile, splitk,
                     workspace, workspace_bytes, nullptr, 1, 0, nullptr, stream);
}

// Up to four independent products in one launch of 64x64-tile workgroups (stages = 2 or 3 ring stages for all of
// them).  Product p is described as for rs_gemm_f32 (unsplit: bias / activation / mulsrc / colsum_part epilogue into
// C) or as for rs_gemm_f32_slabs (splitk > 1: slabs left in its workspace, C and the epilogue fields unused).  All
// operands must allow 16-byte accesses (the condition under which rs_gemm_f32 takes its DMA kernel).  The caller
// orders nothing between the products: they must not read one another's results.
// descriptors -> kernel arguments; *blocks = workgroups of the products (0: nothing to do)
static int group_fill(GemmGroupArgs& a, const rs_gemm_desc* d, int count, int* blocks_out) {
    *blocks_out = 0;
    RS_CHECK_ARG(count >= 0 && count <= GROUP_MAX, "rs_gemm_f32_group: at most %d products", GROUP_MAX);
    if (count == 0) return RS_OK;
    RS_CHECK_ARG(d, "rs_gemm_f32_group: null argument");
    int next = 0, used = 0;
    for (int p = 0; p < count; ++p) {
        const rs_gemm_desc& q = d[p];
        RS_CHECK_ARG(q.M >= 0 && q.N >= 0 && q.K >= 0, "rs_gemm_f32_group: negative size");
        if (q.M == 0 || q.N == 0) continue;
        RS_CHECK_ARG(q.A && q.B, "rs_gemm_f32_group: null operand");
        RS_CHECK_ARG(!(q.a_mc && !q.b_nc), "rs_gemm_f32_group: layout (A m-contiguous, B k-contiguous) is not provided");
        RS_CHECK_ARG(q.lda >= (q.a_mc ? q.M : q.K) && q.ldb >= (q.b_nc ? q.N : q.K), "rs_gemm_f32_group: bad leading dims");
        RS_CHECK_ARG(q.splitk >= 1 && (q.splitk == 1 || q.splitk <= rs_cdiv(q.K, BK)),
                     "rs_gemm_f32_group: splitk %d does not fit K=%d", q.splitk, q.K);
        const bool va = aligned16(q.A) && q.lda % 4 == 0 && (q.a_mc ? q.M % 4 == 0 : q.K % 4 == 0);
        const bool vb = aligned16(q.B) && q.ldb % 4 == 0 && (q.b_nc ? q.N % 4 == 0 : q.K % 4 == 0);
        RS_CHECK_ARG(va && vb, "rs_gemm_f32_group: product %d has operands that do not allow 16-byte accesses", p);
        GemmArgs g{};
        g.A = q.A; g.B = q.B; g.M = q.M; g.N = q.N; g.K = q.K; g.lda = q.lda; g.ldb = q.ldb;
        g.splitk = q.splitk;
        g.kchunk = rs_cdiv(rs_cdiv(q.K, q.splitk), BK) * BK;
        if (g.kchunk == 0) g.kchunk = BK;
        if (q.splitk > 1) {
            RS_CHECK_ARG(q.workspace && q.workspace_bytes >= rs_gemm_workspace_bytes(q.M, q.N, q.splitk),
                         "rs_gemm_f32_group: workspace of product %d too small for splitk=%d", p, q.splitk);
            RS_CHECK_ARG(!(q.drop_p > 0.f), "rs_gemm_f32_group: a dropout mask goes with an unsplit product only");
            g.C = q.workspace; g.ldc = q.N; g.ws = q.workspace; g.defer = 1;
            g.vec_epi = (q.N % 4 == 0 && aligned16(q.workspace)) ? 1 : 0;
        } else {
            RS_CHECK_ARG(q.C && q.ldc >= q.N, "rs_gemm_f32_group: bad output of product %d", p);
            RS_CHECK_ARG(!q.mulsrc || q.ldm >= q.N, "rs_gemm_f32_group: bad ldm");
            RS_CHECK_ARG(q.act >= RS_ACT_NONE && q.act <= RS_ACT_LEAKYRELU, "rs_gemm_f32_group: activation %d not fusable", q.act);
            g.C = q.C; g.ldc = q.ldc; g.bias = q.bias; g.act = q.act; g.mulsrc = q.mulsrc; g.ldm = q.ldm;
            g.mul_act = q.mul_act; g.cs_part = q.colsum_part;
            if (q.drop_p > 0.f) {
                RS_CHECK_ARG(q.drop_p <= 1.f, "rs_gemm_f32_group: dropout probability has to be between 0 and 1, but got %f", (double)q.drop_p);
                g.has_drop = 1;
                g.drop = DropKey{q.drop_p, q.drop_salt, q.drop_seed, q.drop_call_host, q.drop_call_dev};
            }
            g.vec_epi = (q.N % 4 == 0 && aligned16(q.C) && q.ldc % 4 == 0 && (!q.bias || aligned16(q.bias)) &&
                         (!q.mulsrc || (aligned16(q.mulsrc) && q.ldm % 4 == 0))) ? 1 : 0;
        }
        a.g[used] = g;
        a.gx[used] = rs_cdiv(q.N, 64);
        a.gy[used] = rs_cdiv(q.M, 64);
        a.layout[used] = q.a_mc ? 2 : (q.b_nc ? 1 : 0);
        a.first[used] = next;
        const long long blocks = (long long)a.gx[used] * a.gy[used] * q.splitk;
        RS_CHECK_ARG(next + blocks < (1LL << 30), "rs_gemm_f32_group: too many tiles");
        next += (int)((blocks + 7) / 8 * 8);
        ++used;
    }
    if (used == 0) return RS_OK;
    a.count = used;
    a.first[used] = next;
    *blocks_out = next;
    return RS_OK;
}

int rs_gemm_f32_group(const rs_gemm_desc* d, int count, int stages, hipStream_t stream) {
    RS_CHECK_ARG(stages >= 2 && stages <= 4, "rs_gemm_f32_group: 2, 3 or 4 ring stages");
    GemmGroupArgs a{};
    int next = 0;
    const int rc = group_fill(a, d, count, &next);
    if (rc != RS_OK || next == 0) return rc;
    const size_t smem = (size_t)stages * (64 + 64) * 32 * sizeof(float);
    auto kern = stages == 2 ? gemm_f32_group_kernel<2> : stages == 3 ? gemm_f32_group_kernel<3> : gemm_f32_group_kernel<4>;
    static bool attr_set[3] = {false, false, false};
    if (!attr_set[stages - 2]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess) {
            rs_set_error("rs_gemm_f32_group: cannot raise dynamic LDS limit to %zu", smem);
            return RS_ERR_LAUNCH;
        }
        attr_set[stages - 2] = true;
    }
    hipLaunchKernelGGL(kern, dim3(next), dim3(256), smem, stream, a);
    RS_CHECK_LAUNCH("rs_gemm_f32_group");
    return RS_OK;
}

// rs_gemm_f32_group (3 ring stages) with rs_sparse_bwd_sort of a columnar batch idx [F, B] (B <= rs_group_sort_max_entries())
// riding in the same launch: the sort of the NEXT batch beside a block's backward products.  Arguments after `count`
// as for rs_sort_gather.  count may be 0 (then only the sort runs).  phase 0: the whole sort; phase 1: its first
// split_passes radix passes (4 key bits each above 1024 entries per field, 6 below), the items parked in the plan's
// scratch; phase 2 (same split_passes, a later launch): the remaining passes and the plan.  A sort workgroup that
// shares its compute unit with GEMM workgroups runs three to four times as long as alone, so the two halves ride
// in two launches where there are two.
int rs_group_sort_max_entries(void) { return 256 * 16; }

int rs_gemm_f32_group_sort(const rs_gemm_desc* d, int count, const int64_t* idx, const int* sub_off, const int* sub_cnt,
                           const int* pay_base, int cap, int L, const int64_t* vocab, const int64_t* row_offset, int* plan,
                           int B, int F, int phase, int split_passes, hipStream_t stream) {
    RS_CHECK_ARG(idx && sub_off && sub_cnt && pay_base && vocab && row_offset && plan, "rs_gemm_f32_group_sort: null argument");
    RS_CHECK_ARG(B > 0 && F > 0 && L > 0 && cap >= B, "rs_gemm_f32_group_sort: bad sizes");
    RS_CHECK_ARG(B <= 256 * 16, "rs_gemm_f32_group_sort: %d entries per field exceed %d", B, 256 * 16);
    RS_CHECK_ARG(phase >= 0 && phase <= 2 && (phase == 0 || split_passes > 0), "rs_gemm_f32_group_sort: bad phase");
    GemmGroupArgs a{};
    int next = 0;
    const int rc = group_fill(a, d, count, &next);
    if (rc != RS_OK) return rc;
    SortArgs sa{};
    sa.keys = idx; sa.sub_off = sub_off; sa.sub_cnt = sub_cnt; sa.pay_base = pay_base; sa.pay_stride = F;
    sa.nsrc = 1; sa.F = F; sa.cap = cap; sa.L = L; sa.vocab = vocab; sa.row_offset = row_offset;
    const PlanView pv = carve_plan(plan, F, cap);
    sa.sorted_src = pv.sorted_src; sa.entry_chunk = pv.entry_chunk; sa.entry_seg = pv.entry_seg;
    sa.chunk_start = pv.chunk_start; sa.seg_row = pv.seg_row; sa.seg_first_chunk = pv.seg_first_chunk;
    sa.counts = pv.counts;
    sa.phase = phase; sa.split_passes = split_passes;
    sa.spill = reinterpret_cast<unsigned long long*>(plan + (size_t)F * (6 * (size_t)cap + 2 + 4));   // the plan's tile scratch
    const int lead = (F + 7) / 8 * 8;
    const int items = B <= 1024 ? 4 : 16;
    const size_t gsm = (size_t)3 * (64 + 64) * 32 * sizeof(float), ssm = sort_lds(items, 256);
    const size_t smem = gsm > ssm ? gsm : ssm;
    auto kern = items == 4 ? gemm_f32_group_sort_kernel<3, 4> : gemm_f32_group_sort_kernel<3, 16>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[items == 16]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess) {
            rs_set_error("rs_gemm_f32_group_sort: cannot raise dynamic LDS limit to %zu", smem);
            return RS_ERR_LAUNCH;
        }
        attr_set[items == 16] = true;
    }
    hipLaunchKernelGGL(kern, dim3(lead + next), dim3(256), smem, stream, a, sa, lead);
    RS_CHECK_LAUNCH("rs_gemm_f32_group_sort");
    return RS_OK;
}

// C[p][e] = sum_z ws[p][z * total[p] + e], z < splitk[p] in slab order, e < total[p], for p < count (<= 8) products in
// one launch.  All four arrays are HOST arrays of length count (the pointers in them are device pointers).
int rs_splitk_reduce_multi(const float* const* ws, float* const* C, const long long* total, const int* splitk, int count,
                           hipStream_t stream) {
    RS_CHECK_ARG(count >= 0 && count <= SK_MAX, "rs_splitk_reduce_multi: at most %d products", SK_MAX);
    if (count == 0) return RS_OK;
    RS_CHECK_ARG(ws && C && total && splitk, "rs_splitk_reduce_multi: null argument");
    SplitKMultiArgs a;
    const int blocks = splitk_multi_fill(a, ws, C, total, splitk, count);
    RS_CHECK_ARG(blocks >= 0, "rs_splitk_reduce_multi: bad product (null pointer, empty output or splitk < 1)");
    hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3(blocks), dim3(256), 0, stream, a);
    RS_CHECK_LAUNCH("rs_splitk_reduce_multi");
    return RS_OK;
}

// out[p][n] = sum_r part[p][r*ld[p] + n], r < rows[p], n < n[p], for p < count (<= 8) problems in one launch.
// All five arrays are HOST arrays of length count (the pointers in them are device pointers).
int rs_colsum_finish_multi(const float* const* part, const int* rows, const int* n, const int* ld, float* const* out,
                           int count, hipStream_t stream) {
    RS_CHECK_ARG(count >= 0 && count <= CF_MAX, "rs_colsum_finish_multi: at most %d problems", CF_MAX);
    if (count == 0) return RS_OK;
    RS_CHECK_ARG(part && rows && n && ld && out, "rs_colsum_finish_multi: null argument");
    ColsumFinishArgs a;
    const int blocks = colsum_finish_fill(a, part, rows, n, ld, out, count);
    RS_CHECK_ARG(blocks >= 0, "rs_colsum_finish_multi: bad problem (null pointer, negative rows or ld < n)");
    hipLaunchKernelGGL(colsum_finish_multi_kernel, dim3(blocks), dim3(256), 0, stream, a);
    RS_CHECK_LAUNCH("rs_colsum_finish_multi");
    return RS_OK;
}

size_t rs_colsum_workspace_bytes(int M, int N) { return (size_t)rs_cdiv(M, CS_ROWS) * N * sizeof(float); }

// out[n] = sum_m scale[m] * X[m, n]; scale may be null (plain column sums: the bias gradient of
// nn.Linear, model/utils/Layers.py:42, and the dgamma/dbeta sums of LayerNorm).  With a scale vector
// it is the weight gradient of a 1-wide layer.  workspace: rs_colsum_workspace_bytes(M, N).
int rs_colsum_f32(const float* X, int M, int N, int ld, const float* scale, float* out, float* workspace,
                  size_t workspace_bytes, hipStream_t stream) {
    RS_CHECK_ARG(M >= 0 && N >= 0 && ld >= N, "rs_colsum_f32: bad sizes");
    if (N == 0) return RS_OK;
    RS_CHECK_ARG(out, "rs_colsum_f32: null output");
    if (M == 0) {
        const hipError_t e = hipMemsetAsync(out, 0, (size_t)N * sizeof(float), stream);
        if (e != hipSuccess) {
            rs_set_error("rs_colsum_f32_noalloc: hipMemsetAsync of the %d outputs failed: %s", N, hipGetErrorString(e));
            return RS_ERR_LAUNCH;
        }
        return RS_OK;
    }
    RS_CHECK_ARG(X, "rs_colsum_f32: null input");
    RS_CHECK_ARG(workspace && workspace_bytes >= rs_colsum_workspace_bytes(M, N), "rs_colsum_f32: workspace too small");
    const int chunks = rs_cdiv(M, CS_ROWS);
    const bool vec = aligned16(X) && ld % 4 == 0 && N % 4 == 0;
    dim3 grid(rs_cdiv(N, 256), chunks);
    if (vec) hipLaunchKernelGGL(colsum_stage1_kernel<true>, grid, dim3(256), 0, stream, X, M, N, ld, scale, workspace);
    else hipLaunchKernelGGL(colsum_stage1_kernel<false>, grid, dim3(256), 0, stream, X, M, N, ld, scale, workspace);
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3(rs_cdiv(N, 64)), dim3(256), 0, stream, workspace, chunks, N, out);
    RS_CHECK_LAUNCH("rs_colsum_f32");
    return RS_OK;
}

// Stage 1 of rs_colsum_f32 alone: part [rs_colsum_partial_rows(M), N] partial column sums (64 rows of X each) for a
// caller that finishes several such sums in one launch (rs_colsum_finish_multi).
int rs_colsum_partial_rows(int M) { return rs_cdiv(M, CS_ROWS); }
int rs_colsum_partials(const float* X, int M, int N, int ld, const float* scale, float* part, hipStream_t stream) {
    RS_CHECK_ARG(M > 0 && N > 0 && ld >= N, "rs_colsum_partials: bad sizes");
    RS_CHECK_ARG(X && part, "rs_colsum_partials: null argument");
    const bool vec = aligned16(X) && ld % 4 == 0 && N % 4 == 0;
    dim3 grid(rs_cdiv(N, 256), rs_cdiv(M, CS_ROWS));
    if (vec) hipLaunchKernelGGL(colsum_stage1_kernel<true>, grid, dim3(256), 0, stream, X, M, N, ld, scale, part);
    else hipLaunchKernelGGL(colsum_stage1_kernel<false>, grid, dim3(256), 0, stream, X, M, N, ld, scale, part);
    RS_CHECK_LAUNCH("rs_colsum_partials");
    return RS_OK;
}

// ---- the entry point that allocates nothing ---------------------------------------------------
// rs_colsum_f32 for callers that cannot provide (or allocate) a workspace: the stage-1 partial sums go to a device
// array that belongs to the library.  Each GPU holds its own copy of the array at its own address; the address is
// looked up once per device, with that device current, and kept.
constexpr size_t NOALLOC_SCRATCH_FLOATS = (size_t)1 << 20;          // 4 MiB per GPU
constexpr int NOALLOC_MAX_DEVICES = 64;
static __device__ __align__(256) float scratch[NOALLOC_SCRATCH_FLOATS];
static std::atomic<float*> g_scratch_ptr[NOALLOC_MAX_DEVICES];

// device address of `scratch` on the current device (null + error message on failure)
static float* noalloc_scratch() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NOALLOC_MAX_DEVICES) {
        rs_set_error("rs_colsum_f32_noalloc: cannot identify the current device (%d)", dev);
        return nullptr;
    }
    float* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (p) return p;
    const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
        rs_set_error("rs_colsum_f32_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
        return nullptr;
    }
    g_scratch_ptr[dev].store(p, std::memory_order_release);
    return p;
}

size_t rs_colsum_noalloc_max_floats(void) { return NOALLOC_SCRATCH_FLOATS; }

// Looks up the current device's address of the array (what the first rs_colsum_f32_noalloc on a device would do): a
// runtime query that may load the code object on that device, so it belongs in front of a stream capture, not inside.
int rs_colsum_noalloc_prepare(void) { return noalloc_scratch() ? RS_OK : RS_ERR_LAUNCH; }

int rs_colsum_f32_noalloc(const float* X, int M, int N, int ld, const float* scale, float* out, hipStream_t stream) {
    RS_CHECK_ARG(M >= 0 && N >= 0 && ld >= N, "rs_colsum_f32_noalloc: bad sizes");
    if (N == 0) return RS_OK;
    RS_CHECK_ARG(out, "rs_colsum_f32_noalloc: null output");
    if (M == 0) {
        if (hipMemsetAsync(out, 0, (size_t)N * sizeof(float), stream) != hipSuccess) return RS_ERR_LAUNCH;
        return RS_OK;
    }
    RS_CHECK_ARG(X, "rs_colsum_f32_noalloc: null input");
    const int chunks = rs_cdiv(M, CS_ROWS);
    RS_CHECK_ARG((size_t)chunks * (size_t)N <= NOALLOC_SCRATCH_FLOATS,
                 "rs_colsum_f32_noalloc: ceil(M/%d) * N = %zu partial sums exceed the library's %zu-float scratch array; "
                 "use rs_colsum_f32 with a workspace of your own", CS_ROWS, (size_t)chunks * (size_t)N, NOALLOC_SCRATCH_FLOATS);
    float* ws = noalloc_scratch();
    if (!ws) return RS_ERR_LAUNCH;
    // stage 1 writes every partial sum stage 2 reads, so nothing of the previous call's contents has to be cleared
    const bool vec = aligned16(X) && ld % 4 == 0 && N % 4 == 0;
    dim3 grid(rs_cdiv(N, 256), chunks);
    if (vec) hipLaunchKernelGGL(colsum_stage1_kernel<true>, grid, dim3(256), 0, stream, X, M, N, ld, scale, ws);
    else hipLaunchKernelGGL(colsum_stage1_kernel<false>, grid, dim3(256), 0, stream, X, M, N, ld, scale, ws);
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3(rs_cdiv(N, 64)), dim3(256), 0, stream, ws, chunks, N, out);
    RS_CHECK_LAUNCH("rs_colsum_f32_noalloc");
    return RS_OK;
}

}  // extern "C"
