// K3: deterministic sparse embedding backward for gfx950.
//
// The reference trains dense nn.Embedding tables: autograd writes a full (V, d) gradient per
// table in which only the rows named by the batch are non-zero (row 0 = padding_idx never
// receives gradient: model/utils/Layers.py:132), then clips the global norm and steps a dense
// optimizer (model/utils/BaseModel.py:141-150).  Here the same row gradients are produced
// without ever touching the untouched rows:
//
//   1. field_sort_kernel    one workgroup per field: stable LSD radix sort (1 bit per pass) of the
//                           field's indices entirely in LDS, then run-length segmentation: one
//                           "segment" per distinct row, cut into "chunks" of at most L entries.
//   2. chunk_reduce_kernel  a lane group per chunk sums its gradient rows in ascending sample
//                           order (sequential adds: fixed order, no atomics).
//   3. segment_finalize     a lane group per segment adds its chunk sums in chunk order, writes the
//                           row gradient, and accumulates ||g||^2 partials per workgroup.
//   4. sparse_update        SGD / Adam / AdamW / Adagrad on the touched rows only, scaled by the
//                           global clip coefficient (a device scalar, so no host round trip).
//      scatter_rowgrad      alternative consumer: expand into a dense (R, d) gradient for callers
//                           that keep the reference's dense torch optimizers.
#include "common.h"
#include "rowstate_body.h"
#include "colsum_finish.h"
#include "gather_body.h"
#include "splitk_multi.h"
#include "plan.h"
#include "sort_body.h"

namespace {


// ---- two-level form for one list per field (columnar batches) of more than 1024 entries -------------------
// The single-workgroup sort keeps F of the 256 compute units busy and pays one pass per 2..4 key bits once a thread
// holds 8 or 16 items.  Here the list is cut into tiles of TS = 1024*IT consecutive entries and three launches of
// F * tiles workgroups each do the work:
//   1. every tile is sorted by a workgroup of its own (few items per thread: 4..6 key bits per pass) -> tiles64
//   2. every item finds its place in the merged order -- its place in its tile plus, by binary search, the items of
//      every other tile that go before it; equal keys of an earlier tile go first, which keeps the sort stable and
//      the order identical to the single-workgroup sort's -- and is written there -> merged64
//   3. the segmentation runs per range of TS merged positions; a workgroup counts the heads before its range itself.
// tiles64 / merged64: [F, cap] of key << 32 | gradient row.
template <int IT>
__device__ __forceinline__ void tile_sort_body(const SortArgs& a, int f, int tile, unsigned long long* tiles64) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long buf[];   // [1024*IT]
    int* scratch = reinterpret_cast<int*>(buf + SORT_THREADS * IT);
    constexpr int TS = SORT_THREADS * IT;
    const int t = threadIdx.x;
    const int cnt = a.sub_cnt[f], off = a.sub_off[f], pb = a.pay_base[f];
    const long long V = a.vocab[f];
    const int base = tile * TS;
    const int nt = cnt - base < TS ? cnt - base : TS;          // entries of this tile (may be <= 0)
    for (int j = t; j < TS; j += SORT_THREADS) {
        unsigned long long e = ~0ull;
        if (j < nt) {
            long long k = a.keys[off + base + j];
            if (k < 0 || k >= V) k = 0;
            e = ((unsigned long long)(unsigned)k << 32) | (unsigned)(pb + (base + j) * a.pay_stride);
        }
        buf[j] = e;
    }
    __syncthreads();
    if (nt <= 0) return;                                       // uniform over the workgroup
    radix_passes<IT>(buf, scratch, key_bits(V));
    unsigned long long* dst = tiles64 + (long long)f * a.cap + base;
    for (int j = t; j < nt; j += SORT_THREADS) dst[j] = buf[j];
}

// first position in the sorted tile (TS keys, sentinels last) whose key is >= key (UPPER: > key)
template <int TS, bool UPPER>
__device__ __forceinline__ int tile_bound(const unsigned* tile, unsigned key) {
    int lo = 0;
#pragma unroll
    for (int step = TS / 2; step > 0; step >>= 1) {
        const unsigned k = tile[lo + step - 1];
        if (UPPER ? k <= key : k < key) lo += step;
    }
    const unsigned k = tile[lo];
    if (UPPER ? k <= key : k < key) lo += 1;
    return lo;
}

constexpr int SORT_MAX_TILES = 4;

template <int IT>
__device__ __forceinline__ void tile_rank_body(const SortArgs& a, int f, int q, int tiles, const unsigned long long* tiles64,
                                               unsigned long long* merged64) {
    extern __shared__ __attribute__((aligned(16))) unsigned tkeys[];            // [tiles * TS]
    constexpr int TS = SORT_THREADS * IT;
    const int t = threadIdx.x;
    const int n = a.sub_cnt[f];
    if (q * TS >= n) return;                                                     // uniform over the workgroup
    const unsigned long long* g = tiles64 + (long long)f * a.cap;
    unsigned long long own[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int i = q * TS + j * SORT_THREADS + t;
        own[j] = i < n ? g[i] : ~0ull;
    }
    for (int i = t; i < tiles * TS; i += SORT_THREADS) tkeys[i] = i < n ? (unsigned)(g[i] >> 32) : 0xFFFFFFFFu;
    __syncthreads();
    unsigned long long* dst = merged64 + (long long)f * a.cap;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int i = q * TS + j * SORT_THREADS + t;
        if (i < n) {
            const unsigned key = (unsigned)(own[j] >> 32);
            int rank = j * SORT_THREADS + t;
#pragma unroll
            for (int o = 0; o < SORT_MAX_TILES; ++o) {
                if (o == q || o >= tiles || o * TS >= n) continue;
                rank += o < q ? tile_bound<TS, true>(tkeys + o * TS, key) : tile_bound<TS, false>(tkeys + o * TS, key);
            }
            dst[rank] = own[j];
        }
    }
}

// ---- merge of nsrc ascending sub-lists per field (the owner side of the row-sharded exchange receives its keys
// that way): no sort at all.  Workgroup (f, s) places the items of sub-list s: position in its own list plus, by
// binary search, the items of every other list that go before it (equal keys: the earlier list first, as the
// stable sort of the concatenated lists would have it).  The keys of all lists of the field sit in LDS, 4 bytes
// each; tile_tail_kernel then segments the merged list.
constexpr int MERGE_MAX_LISTS = 16;
constexpr int MERGE_MAX_ENTRIES = 32768;

template <bool UPPER>
__device__ __forceinline__ int list_bound(const unsigned* list, int len, unsigned key) {
    int lo = 0;
    while (len > 0) {
        const int half = len >> 1;
        const unsigned k = list[lo + half];
        if (UPPER ? k <= key : k < key) { lo += half + 1; len -= half + 1; }
        else len = half;
    }
    return lo;
}

__global__ __launch_bounds__(SORT_THREADS) void lists_rank_kernel(SortArgs a, int max_entries, unsigned long long* merged64) {
    extern __shared__ __attribute__((aligned(16))) unsigned lkeys[];             // [sum of the field's list lengths]
    __shared__ int lbase[MERGE_MAX_LISTS + 1];
    const int f = blockIdx.x / a.nsrc, s = blockIdx.x % a.nsrc;
    const int t = threadIdx.x;
    if (a.sub_cnt[s * a.F + f] == 0) return;                                     // uniform over the workgroup
    if (t == 0) {
        int acc = 0;
        for (int o = 0; o < a.nsrc; ++o) { lbase[o] = acc; acc += a.sub_cnt[o * a.F + f]; }
        lbase[a.nsrc] = acc;
    }
    __syncthreads();
    if (lbase[a.nsrc] > max_entries) return;             // more entries than the caller sized the LDS for: leave the field alone
    const long long V = a.vocab[f];
    for (int o = 0; o < a.nsrc; ++o) {
        const int cnt = lbase[o + 1] - lbase[o], off = a.sub_off[o * a.F + f];
        for (int j = t; j < cnt; j += SORT_THREADS) {
            long long k = a.keys[off + j];
            if (k < 0 || k >= V) k = 0;
            lkeys[lbase[o] + j] = (unsigned)k;
        }
    }
    __syncthreads();
    const int cnt = lbase[s + 1] - lbase[s];
    const int pb = a.pay_base[s * a.F + f];
    unsigned long long* dst = merged64 + (long long)f * a.cap;
    for (int j = t; j < cnt; j += SORT_THREADS) {
        const unsigned key = lkeys[lbase[s] + j];
        int rank = j;
        for (int o = 0; o < a.nsrc; ++o) {
            if (o == s) continue;
            const int len = lbase[o + 1] - lbase[o];
            rank += o < s ? list_bound<true>(lkeys + lbase[o], len, key) : list_bound<false>(lkeys + lbase[o], len, key);
        }
        dst[rank] = ((unsigned long long)key << 32) | (unsigned)(pb + j * a.pay_stride);
    }
}

template <int IT>
__device__ __forceinline__ void tile_tail_body(const SortArgs& a, int f, int q, int tiles, const unsigned long long* merged64,
                                               int max_entries) {
    extern __shared__ __attribute__((aligned(16))) int tail_scratch[];           // [32] + [1024]
    constexpr int TS = SORT_THREADS * IT;
    const int t = threadIdx.x;
    int n = 0;
    for (int s = 0; s < a.nsrc; ++s) n += a.sub_cnt[s * a.F + f];
    const bool last = q == tiles - 1;
    if (n > max_entries) {
        // the lists hold more entries than the launch was sized for (lists_rank_kernel left the field alone): the field
        // gets an empty plan, and counts[3] = -1 tells the host (SparsePlan.check)
        if (last && t == 0) {
            a.chunk_start[(long long)f * (a.cap + 1)] = 0;
            a.seg_first_chunk[(long long)f * (a.cap + 1)] = 0;
            int* c = a.counts + f * 4;
            c[0] = 0; c[1] = 0; c[2] = 0; c[3] = -1;
        }
        return;
    }
    if (q * TS >= n && !last) return;                                            // uniform over the workgroup
    const unsigned long long* src = merged64 + (long long)f * a.cap;
    const int lim = q * TS < n ? q * TS : n;
    int hc = 0, pads = 0;                                                         // heads | chunk heads << 16 before the range
    for (int i = t; i < lim; i += SORT_THREADS) {
        const unsigned key = (unsigned)(src[i] >> 32);
        const bool live = key != 0u;
        const bool head = live && (i == 0 || (unsigned)(src[i - 1] >> 32) != key);
        const bool chead = live && (head || (i % a.L == 0));
        hc += (head ? 1 : 0) + (chead ? (1 << 16) : 0);
        pads += live ? 0 : 1;
    }
    int prior = 0, prior_pads = 0;
    if (lim > 0) {
        (void)block_excl_scan(hc, tail_scratch, &prior);
        (void)block_excl_scan(pads, tail_scratch, &prior_pads);
        __syncthreads();
    }
    segment_tail<IT>(a, f, src, tail_scratch, n, q * TS, prior, prior_pads, last);
}

template <int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void field_sort_kernel(SortArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sort_buf[];
    field_sort_body<ITEMS>(a, blockIdx.x, sort_buf);
}

template <int IT>
__global__ __launch_bounds__(SORT_THREADS) void tile_sort_kernel(SortArgs a, int tiles, unsigned long long* tiles64) {
    tile_sort_body<IT>(a, blockIdx.x / tiles, blockIdx.x % tiles, tiles64);
}

template <int IT>
__global__ __launch_bounds__(SORT_THREADS) void tile_rank_kernel(SortArgs a, int tiles, const unsigned long long* tiles64,
                                                                 unsigned long long* merged64) {
    tile_rank_body<IT>(a, blockIdx.x / tiles, blockIdx.x % tiles, tiles, tiles64, merged64);
}

template <int IT>
__global__ __launch_bounds__(SORT_THREADS) void tile_tail_kernel(SortArgs a, int tiles, const unsigned long long* merged64,
                                                                 int max_entries) {
    tile_tail_body<IT>(a, blockIdx.x / tiles, blockIdx.x % tiles, tiles, merged64, max_entries);
}

// Level 1 of the two-level sort beside the gather (see sort_gather_kernel).
template <int IT, int VEC>
__global__ __launch_bounds__(SORT_THREADS) void tile_sort_gather_kernel(SortArgs a, int tiles, unsigned long long* tiles64,
                                                                        GatherArgs ga) {
    const int nsort = a.F * tiles;
    if ((int)blockIdx.x < nsort) {
        tile_sort_body<IT>(a, blockIdx.x / tiles, blockIdx.x % tiles, tiles64);
        return;
    }
    gather_fused_body<VEC>(ga, ((int)blockIdx.x - nsort) * (SORT_THREADS / 64) + (threadIdx.x >> 6));
}

// The index sort of a batch and its fused gather (K1) need nothing but the batch, and the sort keeps only F
// workgroups busy: one launch runs the F sort workgroups first and the gather on the workgroups behind them, so the
// gather costs no time of its own.
template <int ITEMS, int VEC>
__global__ __launch_bounds__(SORT_THREADS) void sort_gather_kernel(SortArgs a, GatherArgs ga) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sort_buf[];
    if ((int)blockIdx.x < a.F) {
        field_sort_body<ITEMS>(a, blockIdx.x, sort_buf);
        return;
    }
    gather_fused_body<VEC>(ga, ((int)blockIdx.x - a.F) * (SORT_THREADS / 64) + (threadIdx.x >> 6));
}

// The index sort beside the catch-up of the rows the batch is about to read (rowstate_body.h): both need only the
// batch, the sort keeps F workgroups busy and the replay of skipped Adam steps runs on the workgroups behind them.
template <int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void sort_catchup_kernel(SortArgs a, CatchArgs ca, int per_table) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sort_buf[];
    if ((int)blockIdx.x < a.F) {
        field_sort_body<ITEMS>(a, blockIdx.x, sort_buf);
        return;
    }
    catchup_block<SORT_THREADS>(ca, (int)blockIdx.x - a.F, per_table);
}

template <int IT>
__global__ __launch_bounds__(SORT_THREADS) void tile_sort_catchup_kernel(SortArgs a, int tiles, unsigned long long* tiles64,
                                                                         CatchArgs ca, int per_table) {
    const int nsort = a.F * tiles;
    if ((int)blockIdx.x < nsort) {
        tile_sort_body<IT>(a, blockIdx.x / tiles, blockIdx.x % tiles, tiles64);
        return;
    }
    catchup_block<SORT_THREADS>(ca, (int)blockIdx.x - nsort, per_table);
}

// -------------------------------------------------------------------------------------------
struct ReduceArgs {
    const int* sorted_src; const int* entry_chunk; const int* chunk_start; const int* seg_row;
    const int* seg_first_chunk; const int* counts;
    int F, cap, d, C, L;         // C lanes per entry/segment (d/4 or d); L entries per wave window
    // gradient sources
    const float* gout;           // [*, d] gradient rows, indexed by sorted_src (or null)
    const float* g1src;          // [*]    1-dim gradient per gradient row (or null)
    const float* dfm;            // [B]    dL/d(fm logit) per sample (or null): adds dfm*(S - e)
    const float* S;              // [B, d]
    const float* emb;            // [B*F, d]
    const float* dlr;            // [B]    dL/d(lr logit) per sample (or null): 1-dim gradient
    int Fdiv;                    // sample = gradient-row index / Fdiv (columnar batches)
    int ldg, ld1g;               // floats between consecutive gradient rows of gout / values of g1src (d and 1 when packed)
    float* chunk_sum;            // [F, cap, d]
    float* chunk_sum1;           // [F, cap] (or null)
    float* rowgrad;              // [F, cap, d]
    float* rowgrad1;             // [F, cap] (or null)
    float* norm_partials;        // [gridDim.x * gridDim.y] of the finalize launch
    // route != null: the gradient of segment (f, u) goes to position route[f * cap + u] of rowgrad / rowgrad1 taken as
    // buffers of ldr / ldr1 floats per position (negative position: dropped) instead of to slot f * cap + u
    const int* route; int ldr, ldr1;
};

template <int VEC>
__device__ __forceinline__ void ldv(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else *p = v[0];
}

constexpr int RU = 8;   // chunk sums in flight per lane group (segment_finalize)

// One wave per window of L consecutive sorted entries (L*C <= 64 lanes): every entry's gradient row is
// loaded at once (C lanes per entry), then a segmented inclusive scan over the entries -- a fixed
// shuffle tree keyed by the chunk id -- leaves each chunk's sum in its last entry.  Chunks never
// cross a window (chunk heads fall on every multiple of L), so no carries leave the wave.
template <int VEC>
__device__ __forceinline__ void chunk_reduce_body(const ReduceArgs& a, int f, int bx) {
    const int lane = threadIdx.x & 63;
    const int w = bx * 4 + (threadIdx.x >> 6);
    const int n = a.counts[f * 4 + 0];
    if (w * a.L >= n) return;                              // wave-uniform
    const int e = lane / a.C, cc = lane - e * a.C;
    const int i = w * a.L + e;
    const bool valid = (e < a.L) && (i < n);
    const int d = a.d, co = cc * VEC;
    int cid = -1, src = 0;
    if (valid) {
        cid = a.entry_chunk[(long long)f * a.cap + i];
        src = a.sorted_src[(long long)f * a.cap + i];
    }
    float g[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) g[k] = 0.f;
    float g1 = 0.f;
    if (cid >= 0) {
        const int b = src / a.Fdiv;
        if (a.gout) ldv<VEC>(a.gout + (long long)src * a.ldg + co, g);
        if (a.dfm) {
            const float sdf = a.dfm[b];
            float sv[VEC], ev[VEC];
            ldv<VEC>(a.S + (long long)b * d + co, sv);
            ldv<VEC>(a.emb + (long long)src * d + co, ev);
#pragma unroll
            for (int k = 0; k < VEC; ++k) g[k] += sdf * (sv[k] - ev[k]);
        }
        if (a.g1src) g1 += a.g1src[(long long)src * a.ld1g];
        if (a.dlr) g1 += a.dlr[b];
    }
    for (int o = 1; o < a.L; o <<= 1) {
        const int delta = o * a.C;
        const int ucid = __shfl_up(cid, delta, 64);
        const bool take = (e >= o) && (cid >= 0) && (ucid == cid);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float up = __shfl_up(g[k], delta, 64);
            if (take) g[k] += up;
        }
        const float up1 = __shfl_up(g1, delta, 64);
        if (take) g1 += up1;
    }
    const int ncid = __shfl_down(cid, a.C, 64);
    const bool last = (cid >= 0) && (e == a.L - 1 || i + 1 >= n || ncid != cid);
    if (last) {
        stv<VEC>(a.chunk_sum + ((long long)f * a.cap + cid) * d + co, g);
        if (cc == 0 && a.chunk_sum1) a.chunk_sum1[(long long)f * a.cap + cid] = g1;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void chunk_reduce_kernel(ReduceArgs a) {
    chunk_reduce_body<VEC>(a, blockIdx.y, blockIdx.x);
}

// chunk_reduce for workgroups [0, F*wx) and, behind them, the finishing additions of the partial column sums
// (bias gradients) and of the split-K slabs (weight gradients) the dense backward left: independent work that
// waits for the same predecessors.
template <int VEC>
__global__ __launch_bounds__(256) void chunk_reduce_colsum_kernel(ReduceArgs a, int wx, ColsumFinishArgs cf, int cs_blocks,
                                                                  SplitKMultiArgs sk) {
    __shared__ float red[16][17];
    const int b = blockIdx.x, nsp = a.F * wx;
    if (b < nsp) chunk_reduce_body<VEC>(a, b / wx, b % wx);
    else if (b < nsp + cs_blocks) colsum_finish_body(cf, b - nsp, red);
    else splitk_reduce_multi_body(sk, b - nsp - cs_blocks);       // weight-gradient slabs of the dense backward
}

// Sum of the chunk sums [c0, c1) of field f in chunk order, RU loads in flight (acc / acc1 start at zero).
template <int VEC>
__device__ __forceinline__ void add_chunk_sums(const ReduceArgs& a, int f, int c0, int c1, int cc, float (&acc)[VEC], float& acc1) {
    const int d = a.d, co = cc * VEC;
    for (int c = c0; c < c1; c += RU) {
        float p[RU][VEC], p1[RU];
#pragma unroll
        for (int r = 0; r < RU; ++r) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) p[r][e] = 0.f;
            p1[r] = 0.f;
            if (c + r < c1) {
                ldv<VEC>(a.chunk_sum + ((long long)f * a.cap + c + r) * d + co, p[r]);
                if (cc == 0 && a.chunk_sum1) p1[r] = a.chunk_sum1[(long long)f * a.cap + c + r];
            }
        }
#pragma unroll
        for (int r = 0; r < RU; ++r) {          // chunk order: fixed summation order
            if (c + r < c1) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += p[r][e];
                acc1 += p1[r];
            }
        }
    }
}

// A lane group (C lanes) adds the chunk sums of one distinct row in chunk order.  A row with more than HEAVY_CHUNKS
// chunks (one value in more than HEAVY_CHUNKS * L entries of the batch: small vocabularies, large batches) would keep
// its group busy long after the rest of the launch is done, so the groups of its wave share it: group j adds the j-th
// of 64/C contiguous sub-ranges of its chunks, and the sub-range sums are added in sub-range order -- still a fixed
// order, one that depends only on the plan.  (Needs lane groups that do not straddle waves: 64 % C == 0.)
constexpr int HEAVY_CHUNKS = 48;
template <int VEC>
__device__ __forceinline__ void segment_finalize_body(const ReduceArgs& a, int f, int bx, int part_index, float* red) {
    const int grp = threadIdx.x / a.C, cc = threadIdx.x - grp * a.C;
    const int gpb = 256 / a.C;
    const int u = bx * gpb + grp;
    const bool coop = (64 % a.C) == 0;
    const int lane = threadIdx.x & 63;
    float nsq = 0.f;
    const bool mine = grp < gpb && u < a.counts[f * 4 + 2];
    int c0 = 0, c1 = 0;
    if (mine) {
        const int* sf = a.seg_first_chunk + (long long)f * (a.cap + 1);
        c0 = sf[u]; c1 = sf[u + 1];
    }
    bool heavy = coop && (c1 - c0) > HEAVY_CHUNKS;
    if (coop) {
        // Sharing costs about one memory latency per shared row, one row after the other; the rows of a small vocabulary
        // sit in one wave, and several long rows side by side (one per lane group) are then the faster form.
        const unsigned long long hb = __ballot(heavy && cc == 0);
        if (hb) {                                            // wave-uniform
            int mx = c1 - c0;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
            if (mx <= 24 * (int)__popcll(hb)) heavy = false;
        }
    }
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    float acc1 = 0.f;
    if (mine && !heavy) add_chunk_sums<VEC>(a, f, c0, c1, cc, acc, acc1);
    if (coop && __any(heavy)) {                              // wave-uniform
        const int gw = 64 / a.C, gl = lane / a.C;            // lane groups of this wave, this lane's group among them
        for (int h = 0; h < gw; ++h) {
            const int hl = h * a.C;                          // first lane of group h
            if (!__shfl((int)heavy, hl, 64)) continue;       // wave-uniform
            const int h0 = __shfl(c0, hl, 64), h1 = __shfl(c1, hl, 64);
            const int per = (h1 - h0 + gw - 1) / gw;
            const int s0 = min(h1, h0 + gl * per), s1 = min(h1, s0 + per);
            float part[VEC], part1 = 0.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) part[e] = 0.f;
            add_chunk_sums<VEC>(a, f, s0, s1, cc, part, part1);
            float tot[VEC], tot1 = 0.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) tot[e] = 0.f;
            for (int j = 0; j < gw; ++j) {                   // sub-range order
#pragma unroll
                for (int e = 0; e < VEC; ++e) tot[e] += __shfl(part[e], j * a.C + cc, 64);
                tot1 += __shfl(part1, j * a.C, 64);
            }
            if (gl == h) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = tot[e];
                acc1 = tot1;
            }
        }
    }
    if (mine) {
        const int d = a.d, co = cc * VEC;
        long long pos = (long long)f * a.cap + u, ldr = d, ldr1 = 1;
        if (a.route) { pos = a.route[pos]; ldr = a.ldr; ldr1 = a.ldr1; }
        if (pos >= 0 && a.rowgrad) stv<VEC>(a.rowgrad + pos * ldr + co, acc);
#pragma unroll
        for (int e = 0; e < VEC; ++e) nsq += acc[e] * acc[e];
        if (cc == 0 && a.rowgrad1) {
            if (pos >= 0) a.rowgrad1[pos * ldr1] = acc1;
            nsq += acc1 * acc1;
        }
    }
    const float tot = rs_block_sum<256>(nsq, red);
    if (threadIdx.x == 0) a.norm_partials[part_index] = tot;
}

template <int VEC>
__global__ __launch_bounds__(256) void segment_finalize_kernel(ReduceArgs a) {
    __shared__ float red[4];
    segment_finalize_body<VEC>(a, blockIdx.y, blockIdx.x, blockIdx.y * gridDim.x + blockIdx.x, red);
}

// segment_finalize for workgroups [0, F*gx) and, in the same launch, the sum-of-squares partials of a dense
// buffer (the flat gradient of the dense parameters) for workgroups [F*gx, F*gx + n_dense_blocks): two
// independent pieces of the global gradient norm in one launch.
template <int VEC>
__global__ __launch_bounds__(256) void segment_finalize_sumsq_kernel(ReduceArgs a, int gx, const float* __restrict__ dense,
                                                                     long long dense_n, float* __restrict__ dense_partials,
                                                                     int n_dense_blocks) {
    __shared__ float red[4];
    const int b = blockIdx.x, nsp = a.F * gx;
    if (b < nsp) {
        segment_finalize_body<VEC>(a, b / gx, b % gx, b, red);
        return;
    }
    const int db = b - nsp;
    float sq = 0.f;
    const long long stride = (long long)n_dense_blocks * 256;
    for (long long i = (long long)db * 256 + threadIdx.x; i < dense_n; i += stride) sq += dense[i] * dense[i];
    const float tot = rs_block_sum<256>(sq, red);
    if (threadIdx.x == 0) dense_partials[db] = tot;
}

// -------------------------------------------------------------------------------------------
// Segmented reduction for wide rows (W floats, W % 4 == 0, W <= 1024: one float4 per thread of a 256-thread
// workgroup), used where a lane group per entry does not fit a wave (field-aware FFM rows on the row-sharded path).
// Phase 1: one workgroup per chunk of the plan adds the chunk's gradient rows gout[sorted_src[q], :] in ascending
// order; the sum goes straight to rowgrad when the row has a single chunk, else to chunk_part[f, chunk, :].
// Phase 2 adds the chunk sums of multi-chunk rows in chunk order.  Both leave ||g||^2 partials.
__global__ __launch_bounds__(256) void wide_chunk_reduce_kernel(PlanView pv, int cap, int W, const float* __restrict__ gout,
                                                                float* __restrict__ rowgrad, float* __restrict__ chunk_part,
                                                                float* __restrict__ norm_partials) {
    __shared__ float red[4];
    const int f = blockIdx.y;
    const int nchunk = pv.counts[f * 4 + 1];
    const int* sfc = pv.seg_first_chunk + (long long)f * (cap + 1);
    const int* cst = pv.chunk_start + (long long)f * (cap + 1);
    const int* ssrc = pv.sorted_src + (long long)f * cap;
    const int* eseg = pv.entry_seg + (long long)f * cap;
    const int c = 4 * threadIdx.x;
    const bool has = c < W;
    float nsq = 0.f;
    for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
        const int beg = cst[ch], end = cst[ch + 1];
        const int seg = eseg[beg];
        const bool single = sfc[seg + 1] - sfc[seg] == 1;
        if (has) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = beg; q < end; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(gout + (long long)ssrc[q] * W + c);
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
            float* dst = single ? rowgrad + ((long long)f * cap + seg) * W : chunk_part + ((long long)f * cap + ch) * W;
            *reinterpret_cast<float4*>(dst + c) = acc;
            if (single) nsq += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
        }
    }
    const float tot = rs_block_sum<256>(nsq, red);
    if (threadIdx.x == 0) norm_partials[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void wide_finish_kernel(PlanView pv, int cap, int W, float* __restrict__ rowgrad,
                                                          const float* __restrict__ chunk_part,
                                                          float* __restrict__ norm_partials) {
    __shared__ float red[4];
    const int f = blockIdx.y;
    const int nseg = pv.counts[f * 4 + 2];
    const int* sfc = pv.seg_first_chunk + (long long)f * (cap + 1);
    const int c = 4 * threadIdx.x;
    const bool has = c < W;
    float nsq = 0.f;
    for (int s = blockIdx.x; s < nseg; s += gridDim.x) {
        const int c0 = sfc[s], c1 = sfc[s + 1];
        if (c1 - c0 < 2 || !has) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ch = c0; ch < c1; ++ch) {
            const float4 t = *reinterpret_cast<const float4*>(chunk_part + ((long long)f * cap + ch) * W + c);
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        *reinterpret_cast<float4*>(rowgrad + ((long long)f * cap + s) * W + c) = acc;
        nsq += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
    }
    const float tot = rs_block_sum<256>(nsq, red);
    if (threadIdx.x == 0) norm_partials[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

constexpr int WIDE_CHUNK_GROUPS = 1024, WIDE_FINISH_GROUPS = 64;

// -------------------------------------------------------------------------------------------
// hyper[]: 0 lr, 1 beta1, 2 beta2, 3 eps, 4 weight_decay, 5 step_size (= lr / (1-beta1^t)),
//          6 sqrt(1-beta2^t), 7 reserved.  scal[]: 0 total grad norm, 1 clip coefficient.
enum { OPT_SGD = 0, OPT_ADAM = 1, OPT_ADAMW = 2, OPT_ADAGRAD = 3 };

__device__ __forceinline__ void opt_update(int opt, const float* __restrict__ hy, float g, float& w, float* m,
                                           float* v) {
    const float lr = hy[0];
    if (opt == OPT_SGD) {
        w = w - lr * g;
    } else if (opt == OPT_ADAGRAD) {
        const float s = *m + g * g;          // state sum lives in the `m` slot
        *m = s;
        w = w - lr * (g / (sqrtf(s) + hy[3]));
    } else {
        if (opt == OPT_ADAMW) w = w * (1.f - lr * hy[4]);
        const float b1 = hy[1], b2 = hy[2];
        const float mm = *m + (g - *m) * (1.f - b1);
        const float vv = *v * b2 + (1.f - b2) * g * g;
        *m = mm; *v = vv;
        const float denom = sqrtf(vv) / hy[6] + hy[3];
        w = w - hy[5] * (mm / denom);
    }
}

struct UpdateArgs {
    const int* seg_row; const int* counts;
    const float* rowgrad; const float* rowgrad1;
    int F, cap, d, C, opt;
    float* table; float* m; float* v;          // [R, d]
    float* table1; float* m1; float* v1;       // [R] (or null)
    const float* hyper; const float* scal;     // device scalars
    // per-row step stamps of tables whose skipped Adam steps are replayed on touch (rs_adam_catchup): a stepped row
    // is marked as containing *step optimizer steps (null: no stamps)
    int* last; int* last1; const long long* step;
};

template <int VEC>
__device__ __forceinline__ void sparse_update_body(const UpdateArgs& a, int f, int bx) {
    const int grp = threadIdx.x / a.C, cc = threadIdx.x - grp * a.C;
    const int gpb = 256 / a.C;
    const int u = bx * gpb + grp;
    if (grp >= gpb || u >= a.counts[f * 4 + 2]) return;
    const long long row = a.seg_row[(long long)f * a.cap + u];
    const float coef = a.scal ? a.scal[1] : 1.f;
    const int d = a.d, co = cc * VEC;
    if (a.table) {
        const long long o = row * d + co;
        float g[VEC], w[VEC], mm[VEC], vv[VEC];
        ldv<VEC>(a.rowgrad + ((long long)f * a.cap + u) * d + co, g);
        ldv<VEC>(a.table + o, w);
        if (a.m) ldv<VEC>(a.m + o, mm);
        if (a.v) ldv<VEC>(a.v + o, vv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) opt_update(a.opt, a.hyper, g[e] * coef, w[e], a.m ? &mm[e] : nullptr, a.v ? &vv[e] : nullptr);
        stv<VEC>(a.table + o, w);
        if (a.m) stv<VEC>(a.m + o, mm);
        if (a.v) stv<VEC>(a.v + o, vv);
        if (cc == 0 && a.last) a.last[row] = (int)*a.step;
    }
    if (cc == 0 && a.table1) {
        float w = a.table1[row];
        opt_update(a.opt, a.hyper, a.rowgrad1[(long long)f * a.cap + u] * coef, w, a.m1 ? a.m1 + row : nullptr,
                   a.v1 ? a.v1 + row : nullptr);
        a.table1[row] = w;
        if (a.last1) a.last1[row] = (int)*a.step;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void sparse_update_kernel(UpdateArgs a) {
    sparse_update_body<VEC>(a, blockIdx.y, blockIdx.x);
}

// The touched-rows step for workgroups [0, F*gx) and the flat dense step for the n_dense_blocks behind them: both
// wait for the same clip coefficient and optimizer scalars, so they share one launch.
template <int VEC>
__global__ __launch_bounds__(256) void sparse_dense_update_kernel(UpdateArgs a, int gx, float* __restrict__ w,
                                                                  const float* __restrict__ g, float* m, float* v,
                                                                  long long n, int n_dense_blocks) {
    const int b = blockIdx.x, nsp = a.F * gx;
    if (b < nsp) {
        sparse_update_body<VEC>(a, b / gx, b % gx);
        return;
    }
    const float coef = a.scal ? a.scal[1] : 1.f;
    const long long stride = (long long)n_dense_blocks * 256;
    for (long long i = (long long)(b - nsp) * 256 + threadIdx.x; i < n; i += stride) {
        float ww = w[i];
        opt_update(a.opt, a.hyper, g[i] * coef, ww, m ? m + i : nullptr, v ? v + i : nullptr);
        w[i] = ww;
    }
}

// sparse_dense_update_kernel with, in front of it, the replay of skipped Adam steps for the distinct rows of the NEXT
// batch (its sorted plan; rowstate_body.h, catchup_seg_block): rows the update steps carry the stamp *step and are left
// alone by the replay, every other row of the next batch is not touched by the update -- the two never meet on a row.
template <int VEC>
__global__ __launch_bounds__(256) void sparse_dense_update_ahead_kernel(UpdateArgs a, int gx, float* __restrict__ w,
                                                                        const float* __restrict__ g, float* m, float* v,
                                                                        long long n, int n_dense_blocks, SegCatchArgs sc,
                                                                        int sx) {
    const int nca = sc.F * sx;
    int b = blockIdx.x;
    if (b < nca) {
        catchup_seg_block<VEC, 256>(sc, b / sx, b % sx);
        return;
    }
    b -= nca;
    const int nsp = a.F * gx;
    if (b < nsp) {
        sparse_update_body<VEC>(a, b / gx, b % gx);
        return;
    }
    const float coef = a.scal ? a.scal[1] : 1.f;
    const long long stride = (long long)n_dense_blocks * 256;
    for (long long i = (long long)(b - nsp) * 256 + threadIdx.x; i < n; i += stride) {
        float ww = w[i];
        opt_update(a.opt, a.hyper, g[i] * coef, ww, m ? m + i : nullptr, v ? v + i : nullptr);
        w[i] = ww;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void scatter_rowgrad_kernel(UpdateArgs a) {
    // a.table = dense gradient [R, d] (pre-zeroed), a.table1 = dense 1-dim gradient [R]
    const int f = blockIdx.y;
    const int grp = threadIdx.x / a.C, cc = threadIdx.x - grp * a.C;
    const int gpb = 256 / a.C;
    const int u = blockIdx.x * gpb + grp;
    if (grp >= gpb || u >= a.counts[f * 4 + 2]) return;
    const long long row = a.seg_row[(long long)f * a.cap + u];
    const int d = a.d, co = cc * VEC;
    if (a.table) {
        float g[VEC];
        ldv<VEC>(a.rowgrad + ((long long)f * a.cap + u) * d + co, g);
        stv<VEC>(a.table + row * d + co, g);
    }
    if (cc == 0 && a.table1) a.table1[row] = a.rowgrad1[(long long)f * a.cap + u];
}

// sum over the touched rows of <reduced gradient, stored row> (both tables), one partial per workgroup: what the L2
// term of a table contributes to the squared gradient norm beyond ||g||^2 and lambda^2 ||W||^2
constexpr int DOT_GROUPS = 32;
template <int VEC>
__global__ __launch_bounds__(256) void touched_dot_kernel(UpdateArgs a, float* __restrict__ partials) {
    __shared__ float red[4];
    const int f = blockIdx.y;
    const int grp = threadIdx.x / a.C, cc = threadIdx.x - grp * a.C;
    const int gpb = 256 / a.C;
    const int n = a.counts[f * 4 + 2];
    const int d = a.d, co = cc * VEC;
    float s = 0.f;
    if (grp < gpb) {
        for (int u = blockIdx.x * gpb + grp; u < n; u += gridDim.x * gpb) {
            const long long row = a.seg_row[(long long)f * a.cap + u];
            if (a.table) {
                float g[VEC], w[VEC];
                ldv<VEC>(a.rowgrad + ((long long)f * a.cap + u) * d + co, g);
                ldv<VEC>(a.table + row * d + co, w);
#pragma unroll
                for (int e = 0; e < VEC; ++e) s = fmaf(g[e], w[e], s);
            }
            if (cc == 0 && a.table1) s = fmaf(a.rowgrad1[(long long)f * a.cap + u], a.table1[row], s);
        }
    }
    const float tot = rs_block_sum<256>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

// x *= *scale (a device scalar held in double precision)
__global__ __launch_bounds__(256) void scale_inplace_kernel(float* __restrict__ x, long long n, const double* __restrict__ scale) {
    const float sc = (float)*scale;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= sc;
}

// dense flat-buffer optimizer step (MLP towers, biases): one launch over all dense parameters
__global__ __launch_bounds__(256) void dense_update_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                           float* m, float* v, long long n, int opt,
                                                           const float* __restrict__ hyper,
                                                           const float* __restrict__ scal, float l2) {
    const float coef = scal ? scal[1] : 1.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float ww = w[i];
        opt_update(opt, hyper, g[i] * coef, ww, m ? m + i : nullptr, v ? v + i : nullptr);
        w[i] = ww;
    }
}

// sum of squares -> one partial per workgroup (fixed grid => fixed summation order)
__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ x, long long n,
                                                             float* __restrict__ partials) {
    __shared__ float red[4];
    float s = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += x[i] * x[i];
    const float tot = rs_block_sum<256>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// Sum of n partials over a 1024-thread block: four independent load/add chains per thread, combined in a fixed
// order (deterministic); result valid in every thread.
__device__ __forceinline__ float sum_partials_1024(const float* __restrict__ partials, int n, float* red /* 16 floats */) {
    constexpr int NT = 1024;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = threadIdx.x;
    for (; i + 3 * NT < n; i += 4 * NT) {
        const float a = partials[i], b = partials[i + NT], c = partials[i + 2 * NT], d = partials[i + 3 * NT];
        s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; i < n; i += NT) s0 += partials[i];
    return rs_block_sum<NT>((s0 + s1) + (s2 + s3), red);
}

// total = sqrt(sum partials); coef = min(1, max_norm / (total + 1e-6))   [torch clip_grad_norm_]
__global__ __launch_bounds__(1024) void clip_coef_kernel(const float* __restrict__ partials, int n, float max_norm,
                                                         float* __restrict__ scal) {
    __shared__ float red[16];
    const float tot = sum_partials_1024(partials, n, red);
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(tot);
        scal[0] = nrm;
        float c = max_norm / (nrm + 1e-6f);
        scal[1] = (max_norm > 0.f && c < 1.f) ? c : 1.f;
    }
}

// step += 1; bias corrections in double precision as torch computes them on the host
__global__ void opt_prepare_kernel(long long* step, float* hyper) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long t = *step + 1;
        *step = t;
        const double b1 = (double)hyper[1], b2 = (double)hyper[2];
        const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
        hyper[5] = (float)((double)hyper[0] / bc1);
        hyper[6] = (float)sqrt(bc2);
    }
}

// clip_coef_kernel and opt_prepare_kernel in one launch
__global__ __launch_bounds__(1024) void clip_prepare_kernel(const float* __restrict__ partials, int n, float max_norm,
                                                            float* __restrict__ scal, long long* step, float* hyper) {
    __shared__ float red[16];
    const float tot = sum_partials_1024(partials, n, red);
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(tot);
        scal[0] = nrm;
        float c = max_norm / (nrm + 1e-6f);
        scal[1] = (max_norm > 0.f && c < 1.f) ? c : 1.f;
        const long long t = *step + 1;
        *step = t;
        const double b1 = (double)hyper[1], b2 = (double)hyper[2];
        const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
        hyper[5] = (float)((double)hyper[0] / bc1);
        hyper[6] = (float)sqrt(bc2);
    }
}

// clip_prepare_kernel for SGD with an L2 term lambda/2 ||W||^2 over tables that are stepped on touched rows only and
// kept as w / G.  partials[0, n_sparse): ||g_r||^2 of the touched rows (A); partials[n_sparse, n): the dense part;
// dot[0, n_dot): <g_r, stored row> (times G: B = sum <g_r, w_r>).  state[0] = G, state[1] = S = sum ||w||^2 over all
// rows of all tables.  With every row's gradient g_r + lambda w_r (g_r = 0 off the batch):
//   norm^2 = dense + A + 2 lambda B + lambda^2 S,  c = clip coefficient,  s = 1 - lr c lambda (what a row without
//   data gradient is multiplied by),  G' = G s,  S' = s^2 S - 2 s lr c B + lr^2 c^2 A.
// A touched row's step w' = s w - lr c g is, on the stored row, w'/G' = w/G - (lr c / G') g: the plain SGD step with
// the coefficient scal_tab[1] = c / G'.  gscale = (float)G' is what the next lookup multiplies by.
__global__ __launch_bounds__(1024) void clip_prepare_l2_kernel(const float* __restrict__ partials, int n_sparse, int n,
                                                               const float* __restrict__ dot, int n_dot, float max_norm,
                                                               float lambda, double* __restrict__ state,
                                                               float* __restrict__ scal, float* __restrict__ scal_tab,
                                                               float* __restrict__ gscale, float* loss, long long* step,
                                                               float* hyper) {
    __shared__ float red[16];
    const float A = sum_partials_1024(partials, n_sparse, red);
    __syncthreads();
    const float Dn = sum_partials_1024(partials + n_sparse, n - n_sparse, red);
    __syncthreads();
    const float Bs = sum_partials_1024(dot, n_dot, red);
    if (threadIdx.x == 0) {
        const double G = state[0], S = state[1], lam = (double)lambda, lr = (double)hyper[0];
        const double Bq = G * (double)Bs;
        double n2 = (double)Dn + (double)A + 2.0 * lam * Bq + lam * lam * S;
        if (n2 < 0.0) n2 = 0.0;
        const float nrm = (float)sqrt(n2);
        scal[0] = nrm;
        float c = max_norm / (nrm + 1e-6f);
        c = (max_norm > 0.f && c < 1.f) ? c : 1.f;
        scal[1] = c;
        const double s = 1.0 - lr * (double)c * lam;
        const double Gn = G * s;
        double Sn = s * s * S - 2.0 * s * lr * (double)c * Bq + lr * lr * (double)c * (double)c * (double)A;
        if (Sn < 0.0) Sn = 0.0;
        state[0] = Gn;
        state[1] = Sn;
        scal_tab[0] = nrm;
        scal_tab[1] = (float)((double)c / Gn);
        gscale[0] = (float)Gn;
        if (loss) loss[0] += (float)(0.5 * lam * S);
        const long long t = *step + 1;
        *step = t;
        const double b1 = (double)hyper[1], b2 = (double)hyper[2];
        const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
        hyper[5] = (float)((double)hyper[0] / bc1);
        hyper[6] = (float)sqrt(bc2);
    }
}


int lanes_for(int d, int* VEC, int* C) {
    if (d % 4 == 0 && d / 4 <= 64) { *VEC = 4; *C = d / 4; return 0; }
    if (d <= 64) { *VEC = 1; *C = d; return 0; }
    return -1;
}

// The optimizer and scatter kernels work on one row per lane group of a 256-thread workgroup, so they take
// rows of up to 1024 floats (field-aware FFM rows).
int lanes_wide(int d, int* VEC, int* C) {
    if (d % 4 == 0 && d / 4 <= 256) { *VEC = 4; *C = d / 4; return 0; }
    return lanes_for(d, VEC, C);
}

// entries per wave window for embedding_dim d: the largest power of two L with L*C <= 64
int window_for(int d) {
    int VEC, C;
    if (lanes_for(d, &VEC, &C) != 0) return -1;
    int L = 1;
    while (L * 2 * C <= 64) L *= 2;
    return L;
}

}  // namespace


namespace {
struct TwoLevelShape { int it, tiles; };
// Up to here the three launches cost what the single workgroup per field does (measured, MI355X, 39 fields:
// 4096 entries 24 us either way; 8192: 34 against 48 us; 16384: 58 against 90 us).
constexpr int TWO_LEVEL_ABOVE = 4096;
// tiles of 1024*it entries, at most SORT_MAX_TILES of them
bool two_level_shape(int n, int nsrc, TwoLevelShape* s) {
    if (nsrc != 1 || n <= TWO_LEVEL_ABOVE) return false;
    s->it = n <= 4096 ? 1 : (n <= 8192 ? 2 : 4);
    s->tiles = rs_cdiv(n, SORT_THREADS * s->it);
    return s->tiles <= SORT_MAX_TILES;
}
inline unsigned long long* plan_tiles(int* plan, int F, int cap) {
    return reinterpret_cast<unsigned long long*>(plan + (size_t)F * (6 * (size_t)cap + 2 + 4));
}

// launches 2 and 3 of the two-level sort
int launch_rank_tail(const SortArgs& a, const TwoLevelShape& sh, unsigned long long* tiles64, hipStream_t stream) {
    unsigned long long* merged64 = tiles64 + (size_t)a.F * a.cap;
    const size_t rsm = (size_t)sh.tiles * SORT_THREADS * sh.it * sizeof(unsigned);
    const size_t tsm = (32 + SORT_THREADS) * sizeof(int);
    const dim3 grid(a.F * sh.tiles), block(SORT_THREADS);
#define RS_RANK_TAIL(IT)                                                                            \
    do {                                                                                            \
        static bool attr = false;                                                                   \
        if (!attr && rsm > 48 * 1024) {                                                             \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(tile_rank_kernel<IT>),            \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)rsm) != hipSuccess) { \
                rs_set_error("sort merge: cannot reserve %zu bytes of LDS", rsm);                   \
                return RS_ERR_LAUNCH;                                                               \
            }                                                                                       \
            attr = true;                                                                            \
        }                                                                                           \
        hipLaunchKernelGGL(tile_rank_kernel<IT>, grid, block, rsm, stream, a, sh.tiles, tiles64, merged64); \
        hipLaunchKernelGGL(tile_tail_kernel<IT>, grid, block, tsm, stream, a, sh.tiles, merged64, a.cap); \
    } while (0)
    if (sh.it == 1) RS_RANK_TAIL(1);
    else if (sh.it == 2) RS_RANK_TAIL(2);
    else RS_RANK_TAIL(4);
#undef RS_RANK_TAIL
    return RS_OK;
}
}  // namespace

extern "C" {

// int32 elements of scratch per field-sort output set with capacity `cap` entries per field:
//   sorted_src [F,cap] | entry_chunk [F,cap] | entry_seg [F,cap] | chunk_start [F,cap+1] | seg_row [F,cap] |
//   seg_first_chunk [F,cap+1] | counts [F,4] | sorted tiles and merged list of the two-level sort 2 x [F,cap] x 64 bit
size_t rs_sort_plan_ints(int F, int cap) { return (size_t)F * (6 * (size_t)cap + 2 + 4) + 4 * (size_t)F * cap; }

int rs_sort_max_entries(void) { return SORT_THREADS * 16; }

// Lists of more than 4096 entries given as one sub-list per field (nsrc = 1) are sorted in two levels (tiles, then a
// merge); several sub-lists per field always take the one-workgroup sort.  The plan is the same either way.


// Per-field stable radix sort + segmentation (see file header).  `plan` is int32 scratch of
// rs_sort_plan_ints(F, cap) elements laid out as above.
int rs_sparse_bwd_sort(const int64_t* keys, const int* sub_off, const int* sub_cnt, const int* pay_base,
                       int pay_stride, int nsrc, int F, int cap, int max_entries, int L, const int64_t* vocab,
                       const int64_t* row_offset, int* plan, hipStream_t stream) {
    RS_CHECK_ARG(keys && sub_off && sub_cnt && pay_base && vocab && row_offset && plan, "rs_sparse_bwd_sort: null argument");
    RS_CHECK_ARG(F > 0 && nsrc > 0 && cap > 0 && L > 0, "rs_sparse_bwd_sort: bad sizes");
    RS_CHECK_ARG(max_entries <= cap, "rs_sparse_bwd_sort: max_entries %d exceeds plan capacity %d", max_entries, cap);
    RS_CHECK_ARG(max_entries <= SORT_THREADS * 16,
                 "rs_sparse_bwd_sort: %d entries per field exceed the in-LDS sort limit %d", max_entries,
                 SORT_THREADS * 16);
    SortArgs a{};
    a.keys = keys; a.sub_off = sub_off; a.sub_cnt = sub_cnt; a.pay_base = pay_base; a.pay_stride = pay_stride;
    a.nsrc = nsrc; a.F = F; a.cap = cap; a.L = L; a.vocab = vocab; a.row_offset = row_offset;
    const PlanView pv = carve_plan(plan, F, cap);
    a.sorted_src = pv.sorted_src; a.entry_chunk = pv.entry_chunk; a.entry_seg = pv.entry_seg;
    a.chunk_start = pv.chunk_start; a.seg_row = pv.seg_row; a.seg_first_chunk = pv.seg_first_chunk;
    a.counts = pv.counts;
    TwoLevelShape sh;
    if (two_level_shape(max_entries, nsrc, &sh)) {
        unsigned long long* tiles64 = plan_tiles(plan, F, cap);
        const size_t tsm = sort_lds(sh.it);
        if (sh.it == 1) hipLaunchKernelGGL(tile_sort_kernel<1>, dim3(F * sh.tiles), dim3(SORT_THREADS), tsm, stream, a, sh.tiles, tiles64);
        else if (sh.it == 2) hipLaunchKernelGGL(tile_sort_kernel<2>, dim3(F * sh.tiles), dim3(SORT_THREADS), tsm, stream, a, sh.tiles, tiles64);
        else hipLaunchKernelGGL(tile_sort_kernel<4>, dim3(F * sh.tiles), dim3(SORT_THREADS), tsm, stream, a, sh.tiles, tiles64);
        const int rc = launch_rank_tail(a, sh, tiles64, stream);
        if (rc != RS_OK) return rc;
        RS_CHECK_LAUNCH("rs_sparse_bwd_sort");
        return RS_OK;
    }
    int items = 1;
    while (items * SORT_THREADS < max_entries) items <<= 1;
    const size_t smem = sort_lds(items);   // items, scan scratch, one count per thread
#define RS_SORT_LAUNCH(IT)                                                                          \
    do {                                                                                            \
        static bool attr = false;                                                                   \
        if (!attr) {                                                                                \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(field_sort_kernel<IT>),           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) { \
                rs_set_error("rs_sparse_bwd_sort: cannot reserve %zu bytes of LDS", smem);          \
                return RS_ERR_LAUNCH;                                                               \
            }                                                                                       \
            attr = true;                                                                            \
        }                                                                                           \
        hipLaunchKernelGGL(field_sort_kernel<IT>, dim3(F), dim3(SORT_THREADS), smem, stream, a);   \
    } while (0)
    if (items == 1) RS_SORT_LAUNCH(1);
    else if (items == 2) RS_SORT_LAUNCH(2);
    else if (items == 4) RS_SORT_LAUNCH(4);
    else if (items == 8) RS_SORT_LAUNCH(8);
    else RS_SORT_LAUNCH(16);
#undef RS_SORT_LAUNCH
    RS_CHECK_LAUNCH("rs_sparse_bwd_sort");
    return RS_OK;
}

int rs_merge_max_entries(void) { return MERGE_MAX_ENTRIES; }

// rs_sparse_bwd_sort for sub-lists that are each in ascending key order already: the same plan from a merge (two
// launches of F*nsrc and F*ceil(max_entries/4096) workgroups) instead of a sort.  max_entries (the most entries any
// field has over all its sub-lists) may reach rs_merge_max_entries(); nsrc <= 16.  Sub-lists that are not ascending
// give an unspecified (in-bounds) plan.
int rs_sparse_bwd_merge_lists(const int64_t* keys, const int* sub_off, const int* sub_cnt, const int* pay_base,
                              int pay_stride, int nsrc, int F, int cap, int max_entries, int L, const int64_t* vocab,
                              const int64_t* row_offset, int* plan, hipStream_t stream) {
    RS_CHECK_ARG(keys && sub_off && sub_cnt && pay_base && vocab && row_offset && plan, "rs_sparse_bwd_merge_lists: null argument");
    RS_CHECK_ARG(F > 0 && nsrc > 0 && nsrc <= MERGE_MAX_LISTS && cap > 0 && L > 0, "rs_sparse_bwd_merge_lists: bad sizes");
    RS_CHECK_ARG(max_entries > 0 && max_entries <= cap, "rs_sparse_bwd_merge_lists: max_entries %d exceeds plan capacity %d", max_entries, cap);
    RS_CHECK_ARG(max_entries <= MERGE_MAX_ENTRIES, "rs_sparse_bwd_merge_lists: %d entries per field exceed the limit %d",
                 max_entries, MERGE_MAX_ENTRIES);
    SortArgs a{};
    a.keys = keys; a.sub_off = sub_off; a.sub_cnt = sub_cnt; a.pay_base = pay_base; a.pay_stride = pay_stride;
    a.nsrc = nsrc; a.F = F; a.cap = cap; a.L = L; a.vocab = vocab; a.row_offset = row_offset;
    const PlanView pv = carve_plan(plan, F, cap);
    a.sorted_src = pv.sorted_src; a.entry_chunk = pv.entry_chunk; a.entry_seg = pv.entry_seg;
    a.chunk_start = pv.chunk_start; a.seg_row = pv.seg_row; a.seg_first_chunk = pv.seg_first_chunk;
    a.counts = pv.counts;
    unsigned long long* merged64 = plan_tiles(plan, F, cap) + (size_t)F * cap;
    const size_t rsm = (size_t)max_entries * sizeof(unsigned);
    static size_t reserved = 0;
    if (rsm > 48 * 1024 && rsm > reserved) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lists_rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                MERGE_MAX_ENTRIES * (int)sizeof(unsigned)) != hipSuccess) {
            rs_set_error("rs_sparse_bwd_merge_lists: cannot reserve %d bytes of LDS", MERGE_MAX_ENTRIES * (int)sizeof(unsigned));
            return RS_ERR_LAUNCH;
        }
        reserved = MERGE_MAX_ENTRIES * sizeof(unsigned);
    }
    hipLaunchKernelGGL(lists_rank_kernel, dim3(F * nsrc), dim3(SORT_THREADS), rsm, stream, a, max_entries, merged64);
    const int it = max_entries <= 4096 ? 1 : (max_entries <= 8192 ? 2 : 4);
    const int tiles = rs_cdiv(max_entries, SORT_THREADS * it);
    const size_t tsm = (32 + SORT_THREADS) * sizeof(int);
    const dim3 grid(F * tiles), block(SORT_THREADS);
    if (it == 1) hipLaunchKernelGGL(tile_tail_kernel<1>, grid, block, tsm, stream, a, tiles, merged64, max_entries);
    else if (it == 2) hipLaunchKernelGGL(tile_tail_kernel<2>, grid, block, tsm, stream, a, tiles, merged64, max_entries);
    else hipLaunchKernelGGL(tile_tail_kernel<4>, grid, block, tsm, stream, a, tiles, merged64, max_entries);
    RS_CHECK_LAUNCH("rs_sparse_bwd_merge_lists");
    return RS_OK;
}

// rs_sparse_bwd_sort of a columnar batch idx [F, B] (entry (f, b) names gradient row b*F + f) and
// rs_gather_lr_fm_fwd of the same batch in one launch.  resume_passes > 0: the sort was begun by rs_sort_catchup with
// split_passes = resume_passes on the same plan and batch; this launch runs its remaining passes and writes the plan.
int rs_sort_gather(const int64_t* idx, const int* sub_off, const int* sub_cnt, const int* pay_base, int cap, int L,
                   const int64_t* vocab, const int64_t* row_offset, int* plan, const float* table, const float* table1,
                   const float* bias, float* emb_out, float* sum_out, float* bi_out, float* lr_out, float* fm_out, int B,
                   int F, int d, int* err_flag, int resume_passes, hipStream_t stream) {
    RS_CHECK_ARG(idx && sub_off && sub_cnt && pay_base && vocab && row_offset && plan, "rs_sort_gather: null argument");
    RS_CHECK_ARG(B > 0 && F > 0 && d > 0 && L > 0 && cap >= B, "rs_sort_gather: bad sizes");
    RS_CHECK_ARG(B <= SORT_THREADS * 16, "rs_sort_gather: %d entries per field exceed the in-LDS sort limit %d", B,
                 SORT_THREADS * 16);
    RS_CHECK_ARG(table || table1, "rs_sort_gather: both tables are null");
    RS_CHECK_ARG(table || !(emb_out || sum_out || bi_out || fm_out), "rs_sort_gather: embedding outputs requested without a table");
    RS_CHECK_ARG(table1 || !lr_out, "rs_sort_gather: lr_out requested without table1");
    SortArgs a{};
    a.keys = idx; a.sub_off = sub_off; a.sub_cnt = sub_cnt; a.pay_base = pay_base; a.pay_stride = F;
    a.nsrc = 1; a.F = F; a.cap = cap; a.L = L; a.vocab = vocab; a.row_offset = row_offset;
    const PlanView pv = carve_plan(plan, F, cap);
    a.sorted_src = pv.sorted_src; a.entry_chunk = pv.entry_chunk; a.entry_seg = pv.entry_seg;
    a.chunk_start = pv.chunk_start; a.seg_row = pv.seg_row; a.seg_first_chunk = pv.seg_first_chunk;
    a.counts = pv.counts;
    GatherArgs ga{idx, row_offset, vocab, table, table1, bias, emb_out, sum_out, bi_out, lr_out, fm_out, err_flag, B, F, d, 1, 1};
    int VEC = 1;
    if (!table) { VEC = 1; ga.C = 1; ga.G = 8; ga.d = 1; }
    else RS_CHECK_ARG(pick_layout(d, &VEC, &ga.G, &ga.C) == 0, "rs_sort_gather: unsupported embedding_dim %d", d);
    const int spw = 64 / (ga.G * ga.C);
    const int gblocks = rs_cdiv(rs_cdiv(B, spw), SORT_THREADS / 64);
    TwoLevelShape sh;
    RS_CHECK_ARG(resume_passes >= 0 && (resume_passes == 0 || !two_level_shape(B, 1, &sh)),
                 "rs_sort_gather: a sort of %d entries per field is not cut in two launches", B);
    if (resume_passes > 0) { a.phase = 2; a.split_passes = resume_passes; a.spill = plan_tiles(plan, F, cap); }
    if (two_level_shape(B, 1, &sh)) {
        unsigned long long* tiles64 = plan_tiles(plan, F, cap);
        const size_t tsm = sort_lds(sh.it);
        const dim3 grid(F * sh.tiles + gblocks), block(SORT_THREADS);
#define RS_TSG_LAUNCH(IT, V) hipLaunchKernelGGL((tile_sort_gather_kernel<IT, V>), grid, block, tsm, stream, a, sh.tiles, tiles64, ga)
        if (VEC == 4) {
            if (sh.it == 1) RS_TSG_LAUNCH(1, 4); else if (sh.it == 2) RS_TSG_LAUNCH(2, 4); else RS_TSG_LAUNCH(4, 4);
        } else {
            if (sh.it == 1) RS_TSG_LAUNCH(1, 1); else if (sh.it == 2) RS_TSG_LAUNCH(2, 1); else RS_TSG_LAUNCH(4, 1);
        }
#undef RS_TSG_LAUNCH
        const int rc = launch_rank_tail(a, sh, tiles64, stream);
        if (rc != RS_OK) return rc;
        RS_CHECK_LAUNCH("rs_sort_gather");
        return RS_OK;
    }
    int items = 1;
    while (items * SORT_THREADS < B) items <<= 1;
    const size_t smem = sort_lds(items);
#define RS_SG_LAUNCH(IT, V)                                                                         \
    do {                                                                                            \
        static bool attr = false;                                                                   \
        if (!attr) {                                                                                \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(sort_gather_kernel<IT, V>),       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) { \
                rs_set_error("rs_sort_gather: cannot reserve %zu bytes of LDS", smem);              \
                return RS_ERR_LAUNCH;                                                               \
            }                                                                                       \
            attr = true;                                                                            \
        }                                                                                           \
        hipLaunchKernelGGL((sort_gather_kernel<IT, V>), dim3(F + gblocks), dim3(SORT_THREADS), smem, stream, a, ga); \
    } while (0)
    if (VEC == 4) {
        if (items == 1) RS_SG_LAUNCH(1, 4); else if (items == 2) RS_SG_LAUNCH(2, 4); else if (items == 4) RS_SG_LAUNCH(4, 4);
        else if (items == 8) RS_SG_LAUNCH(8, 4); else RS_SG_LAUNCH(16, 4);
    } else {
        if (items == 1) RS_SG_LAUNCH(1, 1); else if (items == 2) RS_SG_LAUNCH(2, 1); else if (items == 4) RS_SG_LAUNCH(4, 1);
        else if (items == 8) RS_SG_LAUNCH(8, 1); else RS_SG_LAUNCH(16, 1);
    }
#undef RS_SG_LAUNCH
    RS_CHECK_LAUNCH("rs_sort_gather");
    return RS_OK;
}

// rs_sparse_bwd_sort of a columnar batch with rs_adam_catchup (mark_ahead = 0) of the same batch in one launch.
// split_passes > 0 (B <= rs_sort_split_max_entries()): only the first split_passes radix passes run here and the plan
// is NOT written; rs_sort_gather(..., resume_passes = split_passes) on the same plan and batch finishes the sort beside
// the gather, so that neither launch is as long as the whole sort.
int rs_sort_split_max_entries(void) { return TWO_LEVEL_ABOVE; }
int rs_sort_catchup(const int64_t* idx, const int* sub_off, const int* sub_cnt, const int* pay_base, int cap, int L,
                    const int64_t* vocab, const int64_t* row_offset, int* plan, int B, int F, int d, float* table, float* m,
                    float* v, int* last, float* table1, float* m1, float* v1, int* last1, int opt, const long long* step,
                    const float* hyper, const float* bctab, int tab_n, int kmax, int split_passes, hipStream_t stream) {
    RS_CHECK_ARG(idx && sub_off && sub_cnt && pay_base && vocab && row_offset && plan && step && hyper,
                 "rs_sort_catchup: null argument");
    RS_CHECK_ARG(B > 0 && F > 0 && L > 0 && cap >= B, "rs_sort_catchup: bad sizes");
    RS_CHECK_ARG(B <= SORT_THREADS * 16, "rs_sort_catchup: %d entries per field exceed the in-LDS sort limit %d", B,
                 SORT_THREADS * 16);
    RS_CHECK_ARG(opt == RS_OPT_ADAM || opt == RS_OPT_ADAMW, "rs_sort_catchup: optimizer %d has nothing to replay", opt);
    RS_CHECK_ARG(kmax >= 0 && tab_n >= 0 && (tab_n == 0 || bctab), "rs_sort_catchup: bad replay table");
    SortArgs a{};
    a.keys = idx; a.sub_off = sub_off; a.sub_cnt = sub_cnt; a.pay_base = pay_base; a.pay_stride = F;
    a.nsrc = 1; a.F = F; a.cap = cap; a.L = L; a.vocab = vocab; a.row_offset = row_offset;
    const PlanView pv = carve_plan(plan, F, cap);
    a.sorted_src = pv.sorted_src; a.entry_chunk = pv.entry_chunk; a.entry_seg = pv.entry_seg;
    a.chunk_start = pv.chunk_start; a.seg_row = pv.seg_row; a.seg_first_chunk = pv.seg_first_chunk;
    a.counts = pv.counts;
    CatchArgs ca{};
    const int nt = fill_catchup(ca, idx, row_offset, vocab, B, F, d, table, m, v, last, table1, m1, v1, last1, opt, step, 0,
                                hyper, bctab, tab_n, kmax);
    RS_CHECK_ARG(nt > 0, "rs_sort_catchup: no table, or a table without m / v / last, or a row width the kernels do not take (%d)", d);
    const int per_table = rs_cdiv((long long)B * F, SORT_THREADS);
    const int cblocks = per_table * nt;
    TwoLevelShape sh;
    RS_CHECK_ARG(split_passes >= 0 && (split_passes == 0 || !two_level_shape(B, 1, &sh)),
                 "rs_sort_catchup: a sort of %d entries per field is not cut in two launches", B);
    if (split_passes > 0) { a.phase = 1; a.split_passes = split_passes; a.spill = plan_tiles(plan, F, cap); }
    if (two_level_shape(B, 1, &sh)) {
        unsigned long long* tiles64 = plan_tiles(plan, F, cap);
        const size_t tsm = sort_lds(sh.it);
        const dim3 grid(F * sh.tiles + cblocks), block(SORT_THREADS);
#define RS_TSC_LAUNCH(IT) hipLaunchKernelGGL((tile_sort_catchup_kernel<IT>), grid, block, tsm, stream, a, sh.tiles, tiles64, ca, per_table)
        if (sh.it == 1) RS_TSC_LAUNCH(1); else if (sh.it == 2) RS_TSC_LAUNCH(2); else RS_TSC_LAUNCH(4);
#undef RS_TSC_LAUNCH
        const int rc = launch_rank_tail(a, sh, tiles64, stream);
        if (rc != RS_OK) return rc;
        RS_CHECK_LAUNCH("rs_sort_catchup");
        return RS_OK;
    }
    int items = 1;
    while (items * SORT_THREADS < B) items <<= 1;
    const size_t smem = sort_lds(items);
#define RS_SC_LAUNCH(IT)                                                                            \
    do {                                                                                            \
        static bool attr = false;                                                                   \
        if (!attr) {                                                                                \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(sort_catchup_kernel<IT>),         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) { \
                rs_set_error("rs_sort_catchup: cannot reserve %zu bytes of LDS", smem);             \
                return RS_ERR_LAUNCH;                                                               \
            }                                                                                       \
            attr = true;                                                                            \
        }                                                                                           \
        hipLaunchKernelGGL((sort_catchup_kernel<IT>), dim3(F + cblocks), dim3(SORT_THREADS), smem, stream, a, ca, per_table); \
    } while (0)
    if (items == 1) RS_SC_LAUNCH(1); else if (items == 2) RS_SC_LAUNCH(2); else if (items == 4) RS_SC_LAUNCH(4);
    else if (items == 8) RS_SC_LAUNCH(8); else RS_SC_LAUNCH(16);
#undef RS_SC_LAUNCH
    RS_CHECK_LAUNCH("rs_sort_catchup");
    return RS_OK;
}

static int fill_reduce(ReduceArgs& a, const int* plan, int F, int cap, int d, int* VEC) {
    const PlanView pv = carve_plan(const_cast<int*>(plan), F, cap);
    a.sorted_src = pv.sorted_src; a.entry_chunk = pv.entry_chunk; a.chunk_start = pv.chunk_start;
    a.seg_row = pv.seg_row; a.seg_first_chunk = pv.seg_first_chunk; a.counts = pv.counts;
    a.F = F; a.cap = cap; a.d = d; a.ldg = d; a.ld1g = 1;
    return lanes_for(d, VEC, &a.C);
}

// gradient rows that lie ld_gout / ld_g1 floats apart (0: packed); float4 loads need 16-byte aligned rows
static int set_grad_strides(ReduceArgs& a, int VEC, int ld_gout, int ld_g1, const float* gout) {
    if (ld_gout) a.ldg = ld_gout;
    if (ld_g1) a.ld1g = ld_g1;
    if (a.ldg < a.d || a.ld1g < 1) return -1;
    if (VEC == 4 && gout && (a.ldg % 4 != 0 || (reinterpret_cast<uintptr_t>(gout) & 15))) return -1;
    return 0;
}

// Number of norm partials rs_sparse_bwd_reduce writes.
int rs_sparse_bwd_num_partials(int F, int cap, int d) {
    int VEC, C;
    if (lanes_for(d, &VEC, &C) != 0) return -1;
    return F * rs_cdiv(cap, 256 / C);
}

// Segmented reduction of gradient rows into one row gradient per touched table row, plus
// per-workgroup ||g||^2 partials.  Gradient of entry with gradient-row index r (sample b = r/Fdiv):
//   g = gout[r, :] (if gout) + dfm[b] * (S[b, :] - emb[r, :]) (if dfm);  1-dim: g1src[r] + dlr[b].
int rs_sparse_window(int d) { return window_for(d); }

int rs_sparse_bwd_reduce(const int* plan, int F, int cap, int d, const float* gout, const float* g1src,
                         const float* dfm, const float* S, const float* emb, const float* dlr, int Fdiv,
                         float* chunk_sum, float* chunk_sum1, float* rowgrad, float* rowgrad1,
                         float* norm_partials, int ld_gout, int ld_g1, hipStream_t stream) {
    RS_CHECK_ARG(plan && chunk_sum && rowgrad && norm_partials, "rs_sparse_bwd_reduce: null argument");
    RS_CHECK_ARG(!dfm || (S && emb), "rs_sparse_bwd_reduce: dfm needs S and emb");
    RS_CHECK_ARG((chunk_sum1 != nullptr) == (rowgrad1 != nullptr), "rs_sparse_bwd_reduce: chunk_sum1/rowgrad1 mismatch");
    RS_CHECK_ARG(Fdiv > 0, "rs_sparse_bwd_reduce: Fdiv must be positive");
    ReduceArgs a{};
    int VEC;
    RS_CHECK_ARG(fill_reduce(a, plan, F, cap, d, &VEC) == 0, "rs_sparse_bwd_reduce: unsupported d=%d", d);
    a.gout = gout; a.g1src = g1src; a.dfm = dfm; a.S = S; a.emb = emb; a.dlr = dlr; a.Fdiv = Fdiv;
    a.chunk_sum = chunk_sum; a.chunk_sum1 = chunk_sum1; a.rowgrad = rowgrad; a.rowgrad1 = rowgrad1;
    a.norm_partials = norm_partials;
    RS_CHECK_ARG(set_grad_strides(a, VEC, ld_gout, ld_g1, gout) == 0, "rs_sparse_bwd_reduce: bad gradient row strides %d / %d", ld_gout, ld_g1);
    a.L = window_for(d);       // the sort must have been run with the same L (rs_sparse_window(d))
    dim3 grid(rs_cdiv(cap, 256 / a.C), F), block(256);
    dim3 wgrid(rs_cdiv(rs_cdiv(cap, a.L), 4), F);
    if (VEC == 4) {
        hipLaunchKernelGGL(chunk_reduce_kernel<4>, wgrid, block, 0, stream, a);
        hipLaunchKernelGGL(segment_finalize_kernel<4>, grid, block, 0, stream, a);
    } else {
        hipLaunchKernelGGL(chunk_reduce_kernel<1>, wgrid, block, 0, stream, a);
        hipLaunchKernelGGL(segment_finalize_kernel<1>, grid, block, 0, stream, a);
    }
    RS_CHECK_LAUNCH("rs_sparse_bwd_reduce");
    return RS_OK;
}

// rs_sparse_bwd_reduce plus, in the launch of its second phase, the sum-of-squares partials of a dense buffer
// (dense_partials[0 .. n_dense_partials), one per extra workgroup) and, in the launch of its first phase, the
// cs_count column-sum finishes of rs_colsum_finish_multi and the sk_count split-K reductions of rs_splitk_reduce_multi
// (cs_* / sk_* as those entry points' arguments; the counts may be 0).  Their outputs may be part of `dense`: the
// second launch starts after the first has ended.  dense == nullptr with n_dense_partials == 0: no dense buffer, only the
// finishes ride along.
static int reduce_sumsq_impl(const int* plan, int F, int cap, int d, const float* gout, const float* g1src,
                             const float* dfm, const float* S, const float* emb, const float* dlr, int Fdiv,
                             float* chunk_sum, float* chunk_sum1, float* rowgrad, float* rowgrad1, float* norm_partials,
                             const float* dense, long long dense_n, float* dense_partials, int n_dense_partials,
                             const float* const* cs_part, const int* cs_rows, const int* cs_n, const int* cs_ld,
                             float* const* cs_out, int cs_count, const float* const* sk_ws, float* const* sk_out,
                             const long long* sk_total, const int* sk_splitk, int sk_count, int ld_gout, int ld_g1,
                             const int* route, int ld_row, int ld_row1, hipStream_t stream) {
    RS_CHECK_ARG(plan && chunk_sum && (rowgrad || route) && norm_partials, "rs_sparse_bwd_reduce_sumsq: null argument");
    RS_CHECK_ARG(cs_count >= 0 && cs_count <= CF_MAX, "rs_sparse_bwd_reduce_sumsq: at most %d column-sum problems", CF_MAX);
    ColsumFinishArgs cf{};
    int cs_blocks = 0;
    if (cs_count > 0) {
        RS_CHECK_ARG(cs_part && cs_rows && cs_n && cs_ld && cs_out, "rs_sparse_bwd_reduce_sumsq: null column-sum arrays");
        cs_blocks = colsum_finish_fill(cf, cs_part, cs_rows, cs_n, cs_ld, cs_out, cs_count);
        RS_CHECK_ARG(cs_blocks >= 0, "rs_sparse_bwd_reduce_sumsq: bad column-sum problem");
    }
    RS_CHECK_ARG(sk_count >= 0 && sk_count <= SK_MAX, "rs_sparse_bwd_reduce_sumsq: at most %d split-K products", SK_MAX);
    SplitKMultiArgs sk{};
    int sk_blocks = 0;
    if (sk_count > 0) {
        RS_CHECK_ARG(sk_ws && sk_out && sk_total && sk_splitk, "rs_sparse_bwd_reduce_sumsq: null split-K arrays");
        sk_blocks = splitk_multi_fill(sk, sk_ws, sk_out, sk_total, sk_splitk, sk_count);
        RS_CHECK_ARG(sk_blocks >= 0, "rs_sparse_bwd_reduce_sumsq: bad split-K product");
    }
    RS_CHECK_ARG(!dfm || (S && emb), "rs_sparse_bwd_reduce_sumsq: dfm needs S and emb");
    RS_CHECK_ARG((chunk_sum1 != nullptr) == (rowgrad1 != nullptr), "rs_sparse_bwd_reduce_sumsq: chunk_sum1/rowgrad1 mismatch");
    RS_CHECK_ARG(Fdiv > 0, "rs_sparse_bwd_reduce_sumsq: Fdiv must be positive");
    RS_CHECK_ARG((dense && dense_partials && dense_n >= 0 && n_dense_partials > 0) || (!dense && n_dense_partials == 0),
                 "rs_sparse_bwd_reduce_sumsq: bad dense arguments");
    ReduceArgs a{};
    int VEC;
    RS_CHECK_ARG(fill_reduce(a, plan, F, cap, d, &VEC) == 0, "rs_sparse_bwd_reduce_sumsq: unsupported d=%d", d);
    a.gout = gout; a.g1src = g1src; a.dfm = dfm; a.S = S; a.emb = emb; a.dlr = dlr; a.Fdiv = Fdiv;
    a.chunk_sum = chunk_sum; a.chunk_sum1 = chunk_sum1; a.rowgrad = rowgrad; a.rowgrad1 = rowgrad1;
    a.norm_partials = norm_partials;
    RS_CHECK_ARG(set_grad_strides(a, VEC, ld_gout, ld_g1, gout) == 0, "rs_sparse_bwd_reduce_sumsq: bad gradient row strides %d / %d", ld_gout, ld_g1);
    if (route) {
        RS_CHECK_ARG(ld_row >= d && (!rowgrad1 || ld_row1 >= 1) &&
                     (VEC == 1 || !rowgrad || (ld_row % 4 == 0 && (reinterpret_cast<uintptr_t>(rowgrad) & 15) == 0)),
                     "rs_sparse_bwd_reduce_routed: positions of %d / %d floats do not take rows of %d", ld_row, ld_row1, d);
        a.route = route; a.ldr = ld_row; a.ldr1 = ld_row1;
    }
    a.L = window_for(d);
    const int gx = rs_cdiv(cap, 256 / a.C);
    dim3 block(256), wgrid(rs_cdiv(rs_cdiv(cap, a.L), 4), F), fgrid(F * gx + n_dense_partials);
    const int wx = (int)wgrid.x;
    dim3 cgrid(F * wx + cs_blocks + sk_blocks);
    if (VEC == 4) {
        if (cs_blocks + sk_blocks) hipLaunchKernelGGL(chunk_reduce_colsum_kernel<4>, cgrid, block, 0, stream, a, wx, cf, cs_blocks, sk);
        else hipLaunchKernelGGL(chunk_reduce_kernel<4>, wgrid, block, 0, stream, a);
        hipLaunchKernelGGL(segment_finalize_sumsq_kernel<4>, fgrid, block, 0, stream, a, gx, dense, dense_n, dense_partials,
                           n_dense_partials);
    } else {
        if (cs_blocks + sk_blocks) hipLaunchKernelGGL(chunk_reduce_colsum_kernel<1>, cgrid, block, 0, stream, a, wx, cf, cs_blocks, sk);
        else hipLaunchKernelGGL(chunk_reduce_kernel<1>, wgrid, block, 0, stream, a);
        hipLaunchKernelGGL(segment_finalize_sumsq_kernel<1>, fgrid, block, 0, stream, a, gx, dense, dense_n, dense_partials,
                           n_dense_partials);
    }
    RS_CHECK_LAUNCH("rs_sparse_bwd_reduce_sumsq");
    return RS_OK;
}

int rs_sparse_bwd_reduce_sumsq(const int* plan, int F, int cap, int d, const float* gout, const float* g1src,
                               const float* dfm, const float* S, const float* emb, const float* dlr, int Fdiv,
                               float* chunk_sum, float* chunk_sum1, float* rowgrad, float* rowgrad1, float* norm_partials,
                               const float* dense, long long dense_n, float* dense_partials, int n_dense_partials,
                               const float* const* cs_part, const int* cs_rows, const int* cs_n, const int* cs_ld,
                               float* const* cs_out, int cs_count, const float* const* sk_ws, float* const* sk_out,
                               const long long* sk_total, const int* sk_splitk, int sk_count, int ld_gout, int ld_g1,
                               hipStream_t stream) {
    return reduce_sumsq_impl(plan, F, cap, d, gout, g1src, dfm, S, emb, dlr, Fdiv, chunk_sum, chunk_sum1, rowgrad, rowgrad1,
                             norm_partials, dense, dense_n, dense_partials, n_dense_partials, cs_part, cs_rows, cs_n, cs_ld,
                             cs_out, cs_count, sk_ws, sk_out, sk_total, sk_splitk, sk_count, ld_gout, ld_g1, nullptr, 0, 0,
                             stream);
}

// rs_sparse_bwd_reduce_sumsq whose row gradients go where a table of positions says instead of to slot (f, u): segment u
// of field f writes its sum to position route[f * cap + u] of `rows` (ld_row floats per position) and its 1-dim sum to
// the same position of `rows1` (ld_row1 floats per position; may be a column of `rows`); a negative position drops the
// gradient.  The requester side of the row-sharded exchange: one gradient per distinct key straight into the bucket
// positions of the send buffer (what rs_shard_pack copies there from a compact rowgrad).
int rs_sparse_bwd_reduce_routed(const int* plan, int F, int cap, int d, const float* gout, const float* g1src,
                                const float* dfm, const float* S, const float* emb, const float* dlr, int Fdiv,
                                float* chunk_sum, float* chunk_sum1, float* rows, float* rows1, float* norm_partials,
                                const float* dense, long long dense_n, float* dense_partials, int n_dense_partials,
                                const float* const* cs_part, const int* cs_rows, const int* cs_n, const int* cs_ld,
                                float* const* cs_out, int cs_count, const float* const* sk_ws, float* const* sk_out,
                                const long long* sk_total, const int* sk_splitk, int sk_count, int ld_gout, int ld_g1,
                                const int* route, int ld_row, int ld_row1, hipStream_t stream) {
    RS_CHECK_ARG(route, "rs_sparse_bwd_reduce_routed: null route");
    return reduce_sumsq_impl(plan, F, cap, d, gout, g1src, dfm, S, emb, dlr, Fdiv, chunk_sum, chunk_sum1, rows, rows1,
                             norm_partials, dense, dense_n, dense_partials, n_dense_partials, cs_part, cs_rows, cs_n, cs_ld,
                             cs_out, cs_count, sk_ws, sk_out, sk_total, sk_splitk, sk_count, ld_gout, ld_g1, route, ld_row,
                             ld_row1, stream);
}

static int fill_update(UpdateArgs& a, const int* plan, int F, int cap, int d, int* VEC) {
    const PlanView pv = carve_plan(const_cast<int*>(plan), F, cap);
    a.seg_row = pv.seg_row;
    a.counts = pv.counts;
    a.F = F; a.cap = cap; a.d = d;
    return lanes_wide(d, VEC, &a.C);
}

// Wide-row form of rs_sparse_bwd_reduce: gout [*, W] gradient rows indexed by the plan's sorted_src; rowgrad and
// chunk_part (scratch) are [F, cap, W]; norm_partials has rs_sparse_bwd_wide_num_partials(F, cap) floats.
// W % 4 == 0, W <= 1024.
int rs_sparse_bwd_wide_num_partials(int F, int cap) {
    return F * ((cap < WIDE_CHUNK_GROUPS ? cap : WIDE_CHUNK_GROUPS) + (cap < WIDE_FINISH_GROUPS ? cap : WIDE_FINISH_GROUPS));
}

int rs_sparse_bwd_reduce_wide(const int* plan, int F, int cap, int W, const float* gout, float* rowgrad, float* chunk_part,
                              float* norm_partials, hipStream_t stream) {
    RS_CHECK_ARG(plan && gout && rowgrad && chunk_part && norm_partials, "rs_sparse_bwd_reduce_wide: null argument");
    RS_CHECK_ARG(F > 0 && cap > 0 && W > 0 && W % 4 == 0 && W <= 1024, "rs_sparse_bwd_reduce_wide: row width %d not in 4..1024 step 4", W);
    RS_CHECK_ARG(((reinterpret_cast<uintptr_t>(gout) | reinterpret_cast<uintptr_t>(rowgrad) |
                   reinterpret_cast<uintptr_t>(chunk_part)) & 15) == 0, "rs_sparse_bwd_reduce_wide: buffers must be 16-byte aligned");
    const PlanView pv = carve_plan(const_cast<int*>(plan), F, cap);
    const int g1 = cap < WIDE_CHUNK_GROUPS ? cap : WIDE_CHUNK_GROUPS, g2 = cap < WIDE_FINISH_GROUPS ? cap : WIDE_FINISH_GROUPS;
    hipLaunchKernelGGL(wide_chunk_reduce_kernel, dim3(g1, F), dim3(256), 0, stream, pv, cap, W, gout, rowgrad, chunk_part,
                       norm_partials);
    hipLaunchKernelGGL(wide_finish_kernel, dim3(g2, F), dim3(256), 0, stream, pv, cap, W, rowgrad, chunk_part,
                       norm_partials + (size_t)F * g1);
    RS_CHECK_LAUNCH("rs_sparse_bwd_reduce_wide");
    return RS_OK;
}

// Optimizer step on the touched rows only.  opt: 0 SGD, 1 Adam, 2 AdamW, 3 Adagrad (state in m).
int rs_sparse_update(const int* plan, int F, int cap, int d, int opt, const float* rowgrad, const float* rowgrad1,
                     float* table, float* m, float* v, float* table1, float* m1, float* v1,
                     const float* hyper, const float* scal, int* last, int* last1, const long long* step,
                     hipStream_t stream) {
    RS_CHECK_ARG(plan && hyper, "rs_sparse_update: null argument");
    RS_CHECK_ARG(!(last || last1) || step, "rs_sparse_update: row stamps need the step counter");
    RS_CHECK_ARG(opt >= 0 && opt <= 3, "rs_sparse_update: unknown optimizer %d", opt);
    RS_CHECK_ARG(!table || rowgrad, "rs_sparse_update: table without rowgrad");
    RS_CHECK_ARG(!table1 || rowgrad1, "rs_sparse_update: table1 without rowgrad1");
    if (opt == OPT_ADAM || opt == OPT_ADAMW)
        RS_CHECK_ARG((!table || (m && v)) && (!table1 || (m1 && v1)), "rs_sparse_update: Adam needs m and v");
    if (opt == OPT_ADAGRAD) RS_CHECK_ARG((!table || m) && (!table1 || m1), "rs_sparse_update: Adagrad needs the sum state in m");
    UpdateArgs a{};
    int VEC;
    RS_CHECK_ARG(fill_update(a, plan, F, cap, d, &VEC) == 0, "rs_sparse_update: unsupported d=%d", d);
    a.rowgrad = rowgrad; a.rowgrad1 = rowgrad1; a.opt = opt; a.table = table; a.m = m; a.v = v;
    a.table1 = table1; a.m1 = m1; a.v1 = v1; a.hyper = hyper; a.scal = scal;
    a.last = table ? last : nullptr; a.last1 = table1 ? last1 : nullptr; a.step = step;
    dim3 grid(rs_cdiv(cap, 256 / a.C), F), block(256);
    if (VEC == 4) hipLaunchKernelGGL(sparse_update_kernel<4>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(sparse_update_kernel<1>, grid, block, 0, stream, a);
    RS_CHECK_LAUNCH("rs_sparse_update");
    return RS_OK;
}

// rs_sparse_update and rs_dense_update (flat buffer dense_w / dense_g / dense_m / dense_v of dense_n floats) in one launch.
int rs_sparse_dense_update(const int* plan, int F, int cap, int d, int opt, const float* rowgrad, const float* rowgrad1,
                           float* table, float* m, float* v, float* table1, float* m1, float* v1, float* dense_w,
                           const float* dense_g, float* dense_m, float* dense_v, long long dense_n, const float* hyper,
                           const float* scal, int* last, int* last1, const long long* step, hipStream_t stream) {
    RS_CHECK_ARG(plan && hyper && dense_w && dense_g && dense_n >= 0, "rs_sparse_dense_update: null argument");
    RS_CHECK_ARG(!(last || last1) || step, "rs_sparse_dense_update: row stamps need the step counter");
    RS_CHECK_ARG(opt >= 0 && opt <= 3, "rs_sparse_dense_update: unknown optimizer %d", opt);
    RS_CHECK_ARG(!table || rowgrad, "rs_sparse_dense_update: table without rowgrad");
    RS_CHECK_ARG(!table1 || rowgrad1, "rs_sparse_dense_update: table1 without rowgrad1");
    if (opt == OPT_ADAM || opt == OPT_ADAMW)
        RS_CHECK_ARG((!table || (m && v)) && (!table1 || (m1 && v1)) && dense_m && dense_v, "rs_sparse_dense_update: Adam needs m and v");
    if (opt == OPT_ADAGRAD)
        RS_CHECK_ARG((!table || m) && (!table1 || m1) && dense_m, "rs_sparse_dense_update: Adagrad needs the sum state in m");
    UpdateArgs a{};
    int VEC;
    RS_CHECK_ARG(fill_update(a, plan, F, cap, d, &VEC) == 0, "rs_sparse_dense_update: unsupported d=%d", d);
    a.rowgrad = rowgrad; a.rowgrad1 = rowgrad1; a.opt = opt; a.table = table; a.m = m; a.v = v;
    a.table1 = table1; a.m1 = m1; a.v1 = v1; a.hyper = hyper; a.scal = scal;
    a.last = table ? last : nullptr; a.last1 = table1 ? last1 : nullptr; a.step = step;
    const int gx = rs_cdiv(cap, 256 / a.C);
    long long db = (dense_n + 255) / 256;
    const int ndb = (int)(db > 2048 ? 2048 : (db < 1 ? 1 : db));
    dim3 grid(F * gx + ndb), block(256);
    if (VEC == 4) hipLaunchKernelGGL(sparse_dense_update_kernel<4>, grid, block, 0, stream, a, gx, dense_w, dense_g, dense_m,
                                     dense_v, dense_n, ndb);
    else hipLaunchKernelGGL(sparse_dense_update_kernel<1>, grid, block, 0, stream, a, gx, dense_w, dense_g, dense_m, dense_v,
                            dense_n, ndb);
    RS_CHECK_LAUNCH("rs_sparse_dense_update");
    return RS_OK;
}

// rs_sparse_dense_update (which gets no stamps to write) and, in the same launch, rs_adam_catchup_plan(mode 0) for
// `next_plan`, the sorted plan (same F and cap) of the batch the NEXT step trains on: what that step would otherwise do
// before its lookup happens here, beside the update, on rows the update does not touch.  The rows of `plan` must carry
// the stamp *step (an earlier rs_adam_catchup_plan / this entry point promised them to this update).
int rs_sparse_dense_update_ahead(const int* plan, int F, int cap, int d, int opt, const float* rowgrad, const float* rowgrad1,
                                 float* table, float* m, float* v, float* table1, float* m1, float* v1, float* dense_w,
                                 const float* dense_g, float* dense_m, float* dense_v, long long dense_n, const float* hyper,
                                 const float* scal, const int* next_plan, int* last, int* last1, const long long* step,
                                 const float* bctab, int tab_n, int kmax, hipStream_t stream) {
    RS_CHECK_ARG(plan && hyper && dense_w && dense_g && dense_n >= 0 && next_plan && step,
                 "rs_sparse_dense_update_ahead: null argument");
    RS_CHECK_ARG(opt == OPT_ADAM || opt == OPT_ADAMW, "rs_sparse_dense_update_ahead: optimizer %d has nothing to replay", opt);
    RS_CHECK_ARG(!table || rowgrad, "rs_sparse_dense_update_ahead: table without rowgrad");
    RS_CHECK_ARG(!table1 || rowgrad1, "rs_sparse_dense_update_ahead: table1 without rowgrad1");
    RS_CHECK_ARG((!table || (m && v)) && (!table1 || (m1 && v1)) && dense_m && dense_v,
                 "rs_sparse_dense_update_ahead: Adam needs m and v");
    RS_CHECK_ARG(kmax >= 0 && tab_n >= 0 && (tab_n == 0 || bctab), "rs_sparse_dense_update_ahead: bad replay table");
    UpdateArgs a{};
    int VEC;
    RS_CHECK_ARG(fill_update(a, plan, F, cap, d, &VEC) == 0, "rs_sparse_dense_update_ahead: unsupported d=%d", d);
    a.rowgrad = rowgrad; a.rowgrad1 = rowgrad1; a.opt = opt; a.table = table; a.m = m; a.v = v;
    a.table1 = table1; a.m1 = m1; a.v1 = v1; a.hyper = hyper; a.scal = scal;
    a.last = nullptr; a.last1 = nullptr; a.step = step;
    const PlanView nv = carve_plan(const_cast<int*>(next_plan), F, cap);
    SegCatchArgs sc{};
    const int nt = fill_seg_catchup(sc, nv.seg_row, nv.counts, F, cap, d, table, m, v, last, table1, m1, v1, last1, opt, step,
                                    hyper, bctab, tab_n, kmax, 0);
    RS_CHECK_ARG(nt > 0, "rs_sparse_dense_update_ahead: no table, or a table without m / v / last, or a row width the kernels do not take (%d)", d);
    // the replay kernel's lanes per row follow the first table it is given, as the update's do
    const int vec_c = sc.t[0].d % 4 == 0 ? 4 : 1;
    RS_CHECK_ARG(vec_c == VEC, "rs_sparse_dense_update_ahead: row width %d of the replayed table does not fit d=%d", sc.t[0].d, d);
    const int gx = rs_cdiv(cap, 256 / a.C), sx = seg_catchup_blocks(sc, 256);
    long long db = (dense_n + 255) / 256;
    const int ndb = (int)(db > 2048 ? 2048 : (db < 1 ? 1 : db));
    dim3 grid(F * sx + F * gx + ndb), block(256);
    if (VEC == 4) hipLaunchKernelGGL(sparse_dense_update_ahead_kernel<4>, grid, block, 0, stream, a, gx, dense_w, dense_g,
                                     dense_m, dense_v, dense_n, ndb, sc, sx);
    else hipLaunchKernelGGL(sparse_dense_update_ahead_kernel<1>, grid, block, 0, stream, a, gx, dense_w, dense_g, dense_m,
                            dense_v, dense_n, ndb, sc, sx);
    RS_CHECK_LAUNCH("rs_sparse_dense_update_ahead");
    return RS_OK;
}

// Expand row gradients into dense gradients (which the caller zeroed): grad[row, :] = rowgrad.
int rs_sparse_scatter_dense(const int* plan, int F, int cap, int d, const float* rowgrad, const float* rowgrad1,
                            float* grad, float* grad1, hipStream_t stream) {
    RS_CHECK_ARG(plan, "rs_sparse_scatter_dense: null plan");
    RS_CHECK_ARG(!grad || rowgrad, "rs_sparse_scatter_dense: grad without rowgrad");
    RS_CHECK_ARG(!grad1 || rowgrad1, "rs_sparse_scatter_dense: grad1 without rowgrad1");
    UpdateArgs a{};
    int VEC;
    RS_CHECK_ARG(fill_update(a, plan, F, cap, d, &VEC) == 0, "rs_sparse_scatter_dense: unsupported d=%d", d);
    a.rowgrad = rowgrad; a.rowgrad1 = rowgrad1; a.table = grad; a.table1 = grad1;
    dim3 grid(rs_cdiv(cap, 256 / a.C), F), block(256);
    if (VEC == 4) hipLaunchKernelGGL(scatter_rowgrad_kernel<4>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(scatter_rowgrad_kernel<1>, grid, block, 0, stream, a);
    RS_CHECK_LAUNCH("rs_sparse_scatter_dense");
    return RS_OK;
}

// Dense optimizer step over a flat parameter buffer.
int rs_dense_update(float* w, const float* g, float* m, float* v, long long n, int opt, const float* hyper,
                    const float* scal, hipStream_t stream) {
    RS_CHECK_ARG(n >= 0, "rs_dense_update: negative n");
    if (n == 0) return RS_OK;
    RS_CHECK_ARG(w && g && hyper, "rs_dense_update: null argument");
    RS_CHECK_ARG(opt >= 0 && opt <= 3, "rs_dense_update: unknown optimizer %d", opt);
    if (opt == OPT_ADAM || opt == OPT_ADAMW) RS_CHECK_ARG(m && v, "rs_dense_update: Adam needs m and v");
    if (opt == OPT_ADAGRAD) RS_CHECK_ARG(m, "rs_dense_update: Adagrad needs the sum state in m");
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(dense_update_kernel, dim3(blocks), dim3(256), 0, stream, w, g, m, v, n, opt, hyper, scal, 0.f);
    RS_CHECK_LAUNCH("rs_dense_update");
    return RS_OK;
}

// partials[0..nblocks) = per-workgroup sums of x^2 (fixed grid => reproducible).
int rs_sumsq_partials(const float* x, long long n, float* partials, int nblocks, hipStream_t stream) {
    RS_CHECK_ARG(partials && nblocks > 0 && n >= 0, "rs_sumsq_partials: bad arguments");
    RS_CHECK_ARG(x || n == 0, "rs_sumsq_partials: null input");
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3(nblocks), dim3(256), 0, stream, x, n, partials);
    RS_CHECK_LAUNCH("rs_sumsq_partials");
    return RS_OK;
}

// scal[0] = sqrt(sum partials), scal[1] = min(1, max_norm / (scal[0] + 1e-6))
// (nn.utils.clip_grad_norm_ as called at model/utils/BaseModel.py:148); max_norm <= 0 disables.
int rs_clip_coef(const float* partials, int n, float max_norm, float* scal, hipStream_t stream) {
    RS_CHECK_ARG(partials && scal && n >= 0, "rs_clip_coef: bad arguments");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1024), 0, stream, partials, n, max_norm, scal);
    RS_CHECK_LAUNCH("rs_clip_coef");
    return RS_OK;
}

// rs_clip_coef followed by rs_opt_prepare, as one launch.
int rs_clip_and_prepare(const float* partials, int n, float max_norm, float* scal, long long* step, float* hyper,
                        hipStream_t stream) {
    RS_CHECK_ARG(partials && scal && step && hyper && n >= 0, "rs_clip_and_prepare: bad arguments");
    hipLaunchKernelGGL(clip_prepare_kernel, dim3(1), dim3(1024), 0, stream, partials, n, max_norm, scal, step, hyper);
    RS_CHECK_LAUNCH("rs_clip_and_prepare");
    return RS_OK;
}

// ---- SGD with an L2 term on tables that are stepped on touched rows only (model/utils/BaseModel.py:214-231: the
// reference adds lambda/2 ||W||^2 of every nn.Embedding to the loss, so every row has the gradient lambda w, is part of
// the clipped norm and shrinks in every step).  The tables are kept as w / G with one device scalar G for all of them.

// Workgroups per field of rs_sparse_touched_dot (its partials: F times that many floats).
int rs_sparse_touched_dot_groups(int cap, int d) {
    int VEC, C;
    if (cap <= 0 || lanes_wide(d, &VEC, &C) != 0) return -1;
    const int gx = rs_cdiv(cap, 256 / C);
    return gx < DOT_GROUPS ? gx : DOT_GROUPS;
}

// partials[F * groups] = per-workgroup sums of <rowgrad[f, u, :], table[seg_row[f, u], :]> (+ rowgrad1 * table1) over
// the distinct rows of the plan.
int rs_sparse_touched_dot(const int* plan, int F, int cap, int d, const float* rowgrad, const float* rowgrad1,
                          const float* table, const float* table1, float* partials, int n_partials, hipStream_t stream) {
    RS_CHECK_ARG(plan && partials, "rs_sparse_touched_dot: null argument");
    RS_CHECK_ARG(!table || rowgrad, "rs_sparse_touched_dot: table without rowgrad");
    RS_CHECK_ARG(!table1 || rowgrad1, "rs_sparse_touched_dot: table1 without rowgrad1");
    UpdateArgs a{};
    int VEC;
    RS_CHECK_ARG(fill_update(a, plan, F, cap, d, &VEC) == 0, "rs_sparse_touched_dot: unsupported d=%d", d);
    const int gx = rs_sparse_touched_dot_groups(cap, d);
    RS_CHECK_ARG(n_partials == F * gx, "rs_sparse_touched_dot: partials must hold %d floats", F * gx);
    a.rowgrad = rowgrad; a.rowgrad1 = rowgrad1;
    a.table = const_cast<float*>(table); a.table1 = const_cast<float*>(table1);
    dim3 grid(gx, F), block(256);
    if (VEC == 4) hipLaunchKernelGGL(touched_dot_kernel<4>, grid, block, 0, stream, a, partials);
    else hipLaunchKernelGGL(touched_dot_kernel<1>, grid, block, 0, stream, a, partials);
    RS_CHECK_LAUNCH("rs_sparse_touched_dot");
    return RS_OK;
}

// rs_clip_and_prepare with the tables' L2 term (see clip_prepare_l2_kernel): state = {G, S} in double precision,
// scal = {norm, c} for the dense parameters, scal_tab = {norm, c / G'} for rs_sparse_update on the stored rows,
// gscale = (float)G' for rs_gather_lr_fm_fwd_scaled; loss (optional) += lambda/2 * S before the step.
int rs_clip_and_prepare_l2(const float* partials, int n_sparse, int n, const float* dot_partials, int n_dot,
                           float max_norm, float emb_l2, double* state, float* scal, float* scal_tab, float* gscale,
                           float* loss, long long* step, float* hyper, hipStream_t stream) {
    RS_CHECK_ARG(partials && dot_partials && state && scal && scal_tab && gscale && step && hyper,
                 "rs_clip_and_prepare_l2: null argument");
    RS_CHECK_ARG(n_sparse >= 0 && n >= n_sparse && n_dot >= 0 && emb_l2 >= 0.f, "rs_clip_and_prepare_l2: bad sizes");
    hipLaunchKernelGGL(clip_prepare_l2_kernel, dim3(1), dim3(1024), 0, stream, partials, n_sparse, n, dot_partials, n_dot,
                       max_norm, emb_l2, state, scal, scal_tab, gscale, loss, step, hyper);
    RS_CHECK_LAUNCH("rs_clip_and_prepare_l2");
    return RS_OK;
}

// x[0, n) *= (float)*scale.
int rs_scale_inplace(float* x, long long n, const double* scale, hipStream_t stream) {
    RS_CHECK_ARG(n >= 0 && scale, "rs_scale_inplace: bad arguments");
    if (n == 0) return RS_OK;
    RS_CHECK_ARG(x, "rs_scale_inplace: null buffer");
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((int)blocks), dim3(256), 0, stream, x, n, scale);
    RS_CHECK_LAUNCH("rs_scale_inplace");
    return RS_OK;
}

// *step += 1 and refresh the Adam bias-correction scalars in hyper[5], hyper[6].
int rs_opt_prepare(long long* step, float* hyper, hipStream_t stream) {
    RS_CHECK_ARG(step && hyper, "rs_opt_prepare: null argument");
    hipLaunchKernelGGL(opt_prepare_kernel, dim3(1), dim3(64), 0, stream, step, hyper);
    RS_CHECK_LAUNCH("rs_opt_prepare");
    return RS_OK;
}

}  // extern "C"
