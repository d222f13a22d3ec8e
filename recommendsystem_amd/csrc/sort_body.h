// Device side of the per-field index sort of the sparse backward (K3, see sparse.hip): stable LSD radix sort of one
// field's entries in LDS by one workgroup, then run-length segmentation into the plan arrays.  Shared by the sort's own
// launches (sparse.hip, 1024 threads per workgroup) and by the launch in which the sort of the NEXT batch rides beside
// a block's backward GEMMs (gemm.hip, 256 threads per workgroup): the thread count NT is a template parameter, the
// resulting plan does not depend on it (a stable sort has one result).
#pragma once
#include "common.h"

namespace {

constexpr int SORT_THREADS = 1024;

struct SortArgs {
    const int64_t* keys;      // index storage (int64, as the reference's .long() batches)
    const int* sub_off;       // [nsrc*F] start of sub-list (s, f) inside keys
    const int* sub_cnt;       // [nsrc*F] entries in sub-list (s, f)
    const int* pay_base;      // [nsrc*F] gradient-row index of the first entry of sub-list (s, f)
    int pay_stride;           // gradient-row index step between consecutive entries
    int nsrc, F, cap, L;
    const int64_t* vocab;     // [F]
    const int64_t* row_offset;// [F]
    int* sorted_src;          // [F, cap]   gradient-row index of the i-th entry in key order
    int* entry_chunk;         // [F, cap]   chunk of the i-th entry in key order (-1: padding entry)
    int* entry_seg;           // [F, cap]   segment (distinct row) of the i-th entry (-1: padding entry)
    int* chunk_start;         // [F, cap+1] first sorted position of each chunk (+ sentinel)
    int* seg_row;             // [F, cap]   stacked-table row of each segment
    int* seg_first_chunk;     // [F, cap+1] first chunk of each segment (+ sentinel)
    int* counts;              // [F, 4]     n, n_chunk, n_seg, n_padding
    // A one-workgroup sort may be cut in two launches (each beside other work that needs only the batch):
    // phase 1 loads the lists, runs the first split_passes radix passes and leaves the items in spill [F, cap];
    // phase 2 picks them up, runs the remaining passes and writes the plan.  phase 0: the whole sort.
    int phase, split_passes;
    unsigned long long* spill;
};

// Exclusive scan of one int per thread over the NT-thread workgroup; *total gets the sum.
template <int NT = SORT_THREADS>
__device__ __forceinline__ int block_excl_scan(int v, int* scratch /*[NT/64 + 1]*/, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();                       // scratch may still be read from a previous call
    if (lane == 63) scratch[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) {
        const int s = scratch[i];
        if (i < w) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

// ---- stable LSD radix sort of the NT*ITEMS items in buf, DB key bits per pass ---------------------
// Items are taken striped (item j of thread t is element j*NT + t), so the 64 lanes of a wave hold a
// contiguous run of elements: one ballot per digit bit gives every lane the mask of the lanes of its run that hold
// the same digit value -- its rank inside the run, and, in the first lane of every such group, the run's count of that
// value; an exclusive scan over the NB x (runs) counts (1024 of them at most, NB * runs / NT per thread) gives the
// bases.  Values a run does not hold keep the zero of a count array cleared during the previous pass (two
// arrays alternate, so clearing needs no barrier of its own).
template <int ITEMS, int NT = SORT_THREADS> constexpr int radix_digit_bits() {
    if (NT == SORT_THREADS) return ITEMS == 1 ? 6 : (ITEMS == 2 ? 5 : (ITEMS == 4 ? 4 : (ITEMS == 8 ? 3 : 2)));
    // 256 threads: 4 * ITEMS runs of 64 elements, up to 1024 counts
    return ITEMS <= 4 ? 6 : (ITEMS == 8 ? 5 : 4);
}
// passes over the key bits [bit_lo, nbits), bit_lo a multiple of the digit width; scratch: [32] + 2 x [1024] ints
template <int ITEMS, int NT = SORT_THREADS>
__device__ __forceinline__ void radix_passes(unsigned long long* buf, int* scratch, int nbits, int bit_lo = 0) {
    constexpr int DB = radix_digit_bits<ITEMS, NT>();
    constexpr int NB = 1 << DB;
    constexpr int WAVES = NT / 64;
    constexpr int G = WAVES * ITEMS;              // runs of 64 elements
    constexpr int NCNT = NB * G;                  // (digit, run) counts
    static_assert(NCNT % NT == 0 && NCNT <= 1024, "a whole number of (digit, run) counts per thread, 1024 at most");
    constexpr int CPT = NCNT / NT;
    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (bit_lo >= nbits) return;
    int which = 0;
#pragma unroll
    for (int k = 0; k < CPT; ++k) scratch[32 + k * NT + t] = 0;
    __syncthreads();
    for (int bit = bit_lo; bit < nbits; bit += DB, which ^= 1) {
        int* cnt = scratch + 32 + which * 1024;                      // [NB][G]
        int* other = scratch + 32 + (which ^ 1) * 1024;
        const int sh = 32 + bit;
        unsigned long long x[ITEMS], mk[ITEMS];
        int dg[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            x[j] = buf[j * NT + t];
            const int dgt = (int)((x[j] >> sh) & (unsigned long long)(NB - 1));
            unsigned long long own = ~0ull;                          // lanes of this run with my digit
#pragma unroll
            for (int k = 0; k < DB; ++k) {
                const unsigned long long mb = __ballot((dgt >> k) & 1);
                own &= ((dgt >> k) & 1) ? mb : ~mb;
            }
            if ((own & lt) == 0ull) cnt[dgt * G + j * WAVES + w] = __popcll(own);
            dg[j] = dgt;
            mk[j] = own;
        }
        __syncthreads();
        if constexpr (CPT == 1) {
            const int v = cnt[t];
            int tot_unused;
            const int ex = block_excl_scan<NT>(v, scratch, &tot_unused);
            cnt[t] = ex;
            other[t] = 0;                                            // the next pass counts into a cleared array
        } else {
            int loc[CPT], sum = 0;
#pragma unroll
            for (int k = 0; k < CPT; ++k) { loc[k] = cnt[t * CPT + k]; sum += loc[k]; }
            int tot_unused;
            int run = block_excl_scan<NT>(sum, scratch, &tot_unused);
#pragma unroll
            for (int k = 0; k < CPT; ++k) { cnt[t * CPT + k] = run; run += loc[k]; other[t * CPT + k] = 0; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int rank = cnt[dg[j] * G + j * WAVES + w] + __popcll(mk[j] & lt);
            buf[rank] = x[j];
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int key_bits(long long vocab) {
    int nbits = 1;
    const unsigned long long vmax = (unsigned long long)(vocab - 1);
    while ((vmax >> nbits) != 0 && nbits < 32) ++nbits;
    return nbits;
}

// ---- segmentation of the n sorted items in buf: heads of runs of equal keys, chunk heads every L sorted
// positions; writes the plan arrays of field f.  Items are taken striped as in the radix passes, so every wave
// holds 64 consecutive sorted positions: ballots give the heads before a lane inside its run of 64, one scan over
// the run totals gives the heads before the run, and every plan array is written with consecutive
// addresses per wave ------------------------------------------------------------------------------------------
// `src` holds the field's sorted list (LDS or global), indexed by sorted position; the workgroup covers positions
// pos0 .. pos0 + NT*ITEMS.  prior / prior_pads: heads | chunk heads << 16 and padding entries before pos0;
// the workgroup with `last` set writes the end marks and the counts.
template <int ITEMS, int NT = SORT_THREADS>
__device__ __forceinline__ void segment_tail(const SortArgs& a, int f, const unsigned long long* src, int* scratch, int n,
                                             int pos0 = 0, int prior = 0, int prior_pads = 0, bool last = true) {
    constexpr int WAVES = NT / 64;
    constexpr int RUNS = WAVES * ITEMS;
    static_assert(RUNS <= NT, "one run total per thread");
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int* run_heads = scratch + 32;                // [RUNS] heads | chunk heads << 16
    int* run_pads = run_heads + RUNS;             // [RUNS]
    unsigned lo[ITEMS], key[ITEMS];
    int st[ITEMS];                                // bit 0 head, bit 1 chunk head, bit 2 padding, bits 8.. / 16..: heads / chunk heads before the lane in its run
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int i = pos0 + j * NT + t;
        const unsigned long long e = i < n ? src[i] : ~0ull;
        lo[j] = (unsigned)(e & 0xFFFFFFFFull);
        key[j] = (unsigned)(e >> 32);
        const bool valid = i < n;
        const bool pad = valid && key[j] == 0u;
        const bool live = valid && !pad;
        const bool head = live && (i == 0 || (unsigned)(src[i - 1] >> 32) != key[j]);
        const bool chead = live && (head || (i % a.L == 0));
        const unsigned long long hb = __ballot(head), cb = __ballot(chead), pb = __ballot(pad);
        if (lane == 0) {
            run_heads[j * WAVES + w] = __popcll(hb) | (__popcll(cb) << 16);
            run_pads[j * WAVES + w] = __popcll(pb);
        }
        st[j] = (head ? 1 : 0) | (chead ? 2 : 0) | (pad ? 4 : 0) | (__popcll(hb & lt) << 8) | (__popcll(cb & lt) << 16);
    }
    __syncthreads();
    const int rh = t < RUNS ? run_heads[t] : 0, rp = t < RUNS ? run_pads[t] : 0;
    int tot, totpad;
    const int ex = block_excl_scan<NT>(rh, scratch, &tot) + prior;
    (void)block_excl_scan<NT>(rp, scratch, &totpad);
    tot += prior;
    totpad += prior_pads;
    __syncthreads();
    if (t < RUNS) run_heads[t] = ex;
    __syncthreads();
    const int n_seg = tot & 0xFFFF, n_chunk = tot >> 16;
    int* sorted_src = a.sorted_src + (long long)f * a.cap;
    int* entry_chunk = a.entry_chunk + (long long)f * a.cap;
    int* entry_seg = a.entry_seg + (long long)f * a.cap;
    int* chunk_start = a.chunk_start + (long long)f * (a.cap + 1);
    int* seg_row = a.seg_row + (long long)f * a.cap;
    int* seg_first_chunk = a.seg_first_chunk + (long long)f * (a.cap + 1);
    const int roff = (int)a.row_offset[f];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int i = pos0 + j * NT + t;
        if (i < n) {
            const int base = run_heads[j * WAVES + w];
            const int seg = (base & 0xFFFF) + ((st[j] >> 8) & 0xFF);      // heads before this position
            const int chk = (base >> 16) + ((st[j] >> 16) & 0xFF);        // chunk heads before this position
            sorted_src[i] = (int)lo[j];
            if (st[j] & 1) {
                seg_row[seg] = roff + (int)key[j];
                seg_first_chunk[seg] = chk;
            }
            if (st[j] & 2) chunk_start[chk] = i;
            const bool pad = (st[j] & 4) != 0;
            entry_chunk[i] = pad ? -1 : chk + ((st[j] >> 1) & 1) - 1;
            entry_seg[i] = pad ? -1 : seg + (st[j] & 1) - 1;
        }
    }
    if (t == 0 && last) {
        chunk_start[n_chunk] = n;
        seg_first_chunk[n_seg] = n_chunk;
        int* c = a.counts + f * 4;
        c[0] = n; c[1] = n_chunk; c[2] = n_seg; c[3] = totpad;
    }
}

// LDS of one sort workgroup: the items, scan scratch, two count arrays
inline size_t sort_lds(int items, int nt = SORT_THREADS) { return (size_t)nt * items * 8 + (32 + 2 * 1024) * sizeof(int); }

// buf: the workgroup's LDS, sort_lds(ITEMS, NT) bytes
template <int ITEMS, int NT = SORT_THREADS>
__device__ __forceinline__ void field_sort_body(const SortArgs& a, int f, unsigned long long* buf) {
    int* scratch = reinterpret_cast<int*>(buf + NT * ITEMS);                   // [32] + 2 x [1024]
    const int t = threadIdx.x;
    constexpr int CAPB = NT * ITEMS;
    const int nbits = key_bits(a.vocab[f]);
    const int split_bit = radix_digit_bits<ITEMS, NT>() * a.split_passes;
    unsigned long long* spill = a.spill + (long long)f * a.cap;

    int n = 0;
    if (a.phase != 2) {
        // ---- load the field's sub-lists into LDS in list order (position = order of arrival) ----
        for (int s = 0; s < a.nsrc; ++s) {
            const int cnt = a.sub_cnt[s * a.F + f];
            const int off = a.sub_off[s * a.F + f];
            const int pb = a.pay_base[s * a.F + f];
            const long long V = a.vocab[f];
            for (int j = t; j < cnt; j += NT) {
                long long k = a.keys[off + j];
                if (k < 0 || k >= V) k = 0;    // out-of-range was flagged by the forward; treat as padding
                buf[n + j] = ((unsigned long long)(unsigned)k << 32) | (unsigned)(pb + j * a.pay_stride);
            }
            n += cnt;
        }
    } else {
        for (int s = 0; s < a.nsrc; ++s) n += a.sub_cnt[s * a.F + f];
        for (int j = t; j < n; j += NT) buf[j] = spill[j];
    }
    for (int j = n + t; j < CAPB; j += NT) buf[j] = ~0ull;   // sentinels sort last
    __syncthreads();
    if (a.phase == 1) {
        radix_passes<ITEMS, NT>(buf, scratch, min(nbits, split_bit));
        for (int j = t; j < n; j += NT) spill[j] = buf[j];    // the sentinels stay behind the n items
        return;
    }
    radix_passes<ITEMS, NT>(buf, scratch, nbits, a.phase == 2 ? split_bit : 0);
    segment_tail<ITEMS, NT>(a, f, buf, scratch, n);
}

}  // namespace
