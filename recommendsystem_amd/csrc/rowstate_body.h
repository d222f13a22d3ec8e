// Device code of the "catch-up on touch" replay of skipped zero-gradient Adam / AdamW steps (see rowstate.hip for
// what it computes and why).  Included by rowstate.hip (stand-alone launches) and sparse.hip (the launch it shares
// with the index sort).
#pragma once
#include "common.h"

namespace {

enum { RS_OPT_ADAM = 1, RS_OPT_ADAMW = 2 };

struct RowTab {
    float* w; float* m; float* v;   // [R, d]
    int* last;                      // [R]
    int d;
};

struct ReplayConsts {
    float lr, b1, b2, eps, wd;
    int opt, kmax, tab_n;
    const float2* bctab;
};

__device__ __forceinline__ ReplayConsts load_consts(const float* __restrict__ hyper, int opt, const float* bctab, int tab_n,
                                                    int kmax) {
    ReplayConsts c;
    c.lr = hyper[0]; c.b1 = hyper[1]; c.b2 = hyper[2]; c.eps = hyper[3]; c.wd = hyper[4];
    c.opt = opt; c.kmax = kmax; c.tab_n = tab_n;
    c.bctab = reinterpret_cast<const float2*>(bctab);
    return c;
}

// {1/(1 - beta1^k), 1/sqrt(1 - beta2^k)}
__device__ __forceinline__ float2 bias_corrections(const ReplayConsts& c, int k) {
    if (k < c.tab_n) return c.bctab[k];
    const double b1 = (double)c.b1, b2 = (double)c.b2;
    return make_float2((float)(1.0 / (1.0 - pow(b1, (double)k))), (float)(1.0 / sqrt(1.0 - pow(b2, (double)k))));
}

__device__ __forceinline__ float pow_int(float b, int n) {
    if (n <= 8) {
        float p = 1.f;
        for (int i = 0; i < n; ++i) p *= b;
        return p;
    }
    return powf(b, (float)n);
}

// (w, m, v) after the zero-gradient optimizer steps s+1 .. s+n.
template <int VEC>
__device__ __forceinline__ void replay_zero_grad_steps(const ReplayConsts& c, float (&w)[VEC], float (&m)[VEC],
                                                       float (&v)[VEC], int s, int n) {
    const bool adamw = c.opt == RS_OPT_ADAMW;
    const float omb1 = 1.f - c.b1;
    const float decay = 1.f - c.lr * c.wd;
    bool moving = false;
#pragma unroll
    for (int e = 0; e < VEC; ++e) moving |= (m[e] != 0.f);
    int done = 0;
    if (moving) {
        float sq[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) sq[e] = sqrtf(v[e]);
        const float rb2 = sqrtf(c.b2);
        const int nloop = n < c.kmax ? n : c.kmax;
        float2 bc_next = bias_corrections(c, s + 1);
        while (done < nloop) {
            ++done;
            const float2 bc = bc_next;
            bc_next = bias_corrections(c, s + done + 1);       // in flight while this step's arithmetic runs
            const float ss = c.lr * bc.x;
            bool live = false;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                m[e] = fmaf(-m[e], omb1, m[e]);            // exp_avg.lerp_(0, 1 - beta1)
                sq[e] *= rb2;                              // sqrt(exp_avg_sq * beta2)
                const float den = fmaf(sq[e], bc.y, c.eps);
                const float term = ss * m[e] * __builtin_amdgcn_rcpf(den);
                if (adamw) w[e] *= decay;
                w[e] -= term;
                live |= fabsf(term) > 7.5e-9f * fabsf(w[e]);
            }
            if (!live) break;
        }
    }
    const int rest = n - done;
    if (rest > 0) {
        const float pm = pow_int(c.b1, rest);
        const float pw = adamw ? pow_int(decay, rest) : 1.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { m[e] *= pm; w[e] *= pw; }
    }
    // v from its value before the replay (the loop tracked its square root only)
    const float pv = pow_int(c.b2, n);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] *= pv;
}

template <int VEC>
__device__ __forceinline__ void row_ld(const float* p, float (&x)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
    } else {
        x[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void row_st(float* p, const float (&x)[VEC]) {
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
    else *p = x[0];
}

template <int VEC>
__device__ __forceinline__ void replay_row_slice(const ReplayConsts& c, const RowTab& t, long long row, int co, int s, int n) {
    const long long o = row * t.d + co;
    float w[VEC], m[VEC], v[VEC];
    row_ld<VEC>(t.w + o, w);
    row_ld<VEC>(t.m + o, m);
    row_ld<VEC>(t.v + o, v);
    replay_zero_grad_steps<VEC>(c, w, m, v, s, n);
    row_st<VEC>(t.w + o, w);
    row_st<VEC>(t.m + o, m);
    row_st<VEC>(t.v + o, v);
}

struct CatchArgs {
    const int64_t* idx; const int64_t* row_offset; const int64_t* vocab;
    int B, F;
    // list form (sub_off != null): idx holds nlists = nsrc * F key lists, list l = s * F + f of field f starting at
    // idx[sub_off[l]] with sub_cnt[l] (<= per_list) keys -- the buckets a row-sharded owner receives, or the all-gathered
    // distinct rows of every rank
    const int* sub_off; const int* sub_cnt; int nlists, per_list;
    RowTab t[2];
    int opt; const long long* step; int mark_ahead;
    const float* hyper; const float* bctab; int tab_n; int kmax;
};

// Phase 1: one thread per (field, sample) entry looks at its row's stamp; a row that is behind is claimed with
// atomicMax(last[row], steps + mark_ahead): exactly one of the entries that name it sees the old value and becomes its
// owner (whichever it is, the result is the same).  Rows that are up to date -- the hot rows of a Zipf batch, stepped
// and stamped by the previous step's update -- cost one load and no atomic.
// Phase 2: the workgroup replays the rows its threads own, d/VEC lanes per row.
// NT threads per workgroup; `block` = index of the workgroup among those that do this work; s_row, s_last: NT ints each.
template <int VEC, int NT>
__device__ __forceinline__ void catchup_body(const CatchArgs& a, const RowTab& t, int block, int* s_n, int* s_row, int* s_last) {
    if (threadIdx.x == 0) *s_n = 0;
    __syncthreads();
    const long long e = (long long)block * NT + threadIdx.x;
    const int c = (int)*a.step;
    const long long total = a.sub_off ? (long long)a.nlists * a.per_list : (long long)a.B * a.F;
    if (e < total) {
        // fields vary fastest: a workgroup sees every field, so the stale rows of the large-vocabulary fields (nearly
        // every entry of theirs) spread evenly over the workgroups instead of filling a few of them
        int f;
        long long key = 0;
        if (a.sub_off) {
            const int l = (int)(e % a.nlists), j = (int)(e / a.nlists);
            f = l % a.F;
            if (j < a.sub_cnt[l]) key = a.idx[(long long)a.sub_off[l] + j];
        } else {
            f = (int)(e % a.F);
            key = a.idx[(long long)f * a.B + e / a.F];
        }
        if (key > 0 && key < a.vocab[f]) {              // row 0 is the padding row: never stepped (Layers.py:132)
            const int row = (int)(a.row_offset[f] + key);
            const int target = c + a.mark_ahead;
            if (__hip_atomic_load(&t.last[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                const int old = atomicMax(&t.last[row], target);
                if (old < c) {
                    const int p = atomicAdd(s_n, 1);
                    s_row[p] = row; s_last[p] = old;
                }
            }
        }
    }
    __syncthreads();
    const int items = *s_n;
    if (items == 0) return;
    const int C = t.d / VEC, gpb = NT / C;
    const int grp = threadIdx.x / C, cc = threadIdx.x - grp * C;
    if (grp >= gpb) return;
    const ReplayConsts k = load_consts(a.hyper, a.opt, a.bctab, a.tab_n, a.kmax);
    for (int i = grp; i < items; i += gpb) replay_row_slice<VEC>(k, t, s_row[i], cc * VEC, s_last[i], c - s_last[i]);
}

// workgroup `block` of the nt * ceil(B*F / NT) that serve a batch: table block / per_table, entries of block % per_table
template <int NT>
__device__ __forceinline__ void catchup_block(const CatchArgs& a, int block, int per_table) {
    __shared__ int s_n, s_row[NT], s_last[NT];
    const RowTab t = a.t[block / per_table];
    if (t.d % 4 == 0) catchup_body<4, NT>(a, t, block % per_table, &s_n, s_row, s_last);
    else catchup_body<1, NT>(a, t, block % per_table, &s_n, s_row, s_last);
}

// ---- the same replay for the DISTINCT rows of a sorted batch (the plan of the sparse backward) -------------------
// Every row appears once in the plan's seg_row lists, so nobody has to be elected: the thread that looks at a row is
// its only writer.  Used for the batch AFTER the one being stepped ("ahead"): with c = *step optimizer steps
// complete once the launch that carries this code has ended (its own update included),
//   last[row] <  c : the row is not part of the batch being stepped (whose rows carry last == c, see below) and is
//                    brought from last[row] to c steps here;
//   then last[row] = c + 1 for every row of the plan: a promise that the next update steps it, which is also what
//                    tells the catch-up after that one that the row is in good hands.  The update therefore leaves the
//                    stamps alone in this mode, and runs beside this code without touching the same rows' moments.
// mode 1 takes the promise back (last[row] > c -> c) when the batch the plan was made for does not come.
// mode 2 is the plain replay for a plan whose rows the NEXT launches read and step with stamps of their own: rows
// below c are brought up to c and stamped c (what rs_adam_catchup does per entry, here once per distinct row).
struct SegCatchArgs {
    const int* seg_row; const int* counts;      // [F, cap] rows of the distinct keys, [F, 4] (n, chunks, segments, padding)
    int F, cap;
    RowTab t[2]; int nt;
    int nb0;                                    // workgroups per field that serve t[0]; those behind them serve t[1]
    int opt; const long long* step;
    const float* hyper; const float* bctab; int tab_n; int kmax;
    int mode;
    // pay != null (mode 2): the row, once up to date, is also written to every position that names it -- the entries of
    // its segment in the plan, sorted_src[f, e] for e in [chunk_start[first chunk], chunk_start[end chunk]) -- of a buffer
    // of ldp floats per position, table t at column pay_col[t]: the owner of a row-sharded table serves the rows it has
    // just replayed without a second pass over them
    float* pay; int ldp; int pay_col[2];
    const int* sorted_src; const int* chunk_start; const int* seg_first_chunk;
};

// rows a workgroup of NT threads serves: one per group of C lanes (C = lanes of a row of width d)
inline int seg_rows_per_block(int d, int NT) {
    const int C = d % 4 == 0 ? d / 4 : d;
    return NT / C > 0 ? NT / C : 1;
}

// workgroups per field of a launch that serves plans of `cap` slots; sets a.nb0
inline int seg_catchup_blocks(SegCatchArgs& a, int NT) {
    a.nb0 = (a.cap + seg_rows_per_block(a.t[0].d, NT) - 1) / seg_rows_per_block(a.t[0].d, NT);
    return a.nb0 + (a.nt > 1 ? (a.cap + seg_rows_per_block(a.t[1].d, NT) - 1) / seg_rows_per_block(a.t[1].d, NT) : 0);
}

// One table: workgroup bx of the ceil(cap / rows per block) that serve field f.  A group of C lanes per distinct row:
// every lane reads the row's stamp and replays its slice (VEC floats), lane 0 writes the new stamp.
template <int VEC, int NT>
__device__ __forceinline__ void catchup_seg_table(const SegCatchArgs& a, int ti, int f, int bx) {
    const RowTab& t = a.t[ti];
    const int C = t.d / VEC;
    const int rpb = NT / C > 0 ? NT / C : 1;
    const int nseg = a.counts[f * 4 + 2];
    if (bx * rpb >= nseg) return;                                   // uniform over the workgroup
    const int grp = threadIdx.x / C, cc = threadIdx.x - grp * C;
    const int u = bx * rpb + grp;
    const bool act = grp < rpb && u < nseg;
    const int c = (int)*a.step;
    int row = 0, l = c;
    if (act) {
        row = a.seg_row[(long long)f * a.cap + u];
        l = t.last[row];
    }
    __syncthreads();                                                // every lane of a row has read its stamp before it changes
    if (!act) return;
    if (a.mode == 1) {
        if (cc == 0 && l > c) t.last[row] = c;
        return;
    }
    const long long o = (long long)row * t.d + cc * VEC;
    float w[VEC];
    if (l < c) {
        const ReplayConsts k = load_consts(a.hyper, a.opt, a.bctab, a.tab_n, a.kmax);
        float m[VEC], v[VEC];
        row_ld<VEC>(t.w + o, w);
        row_ld<VEC>(t.m + o, m);
        row_ld<VEC>(t.v + o, v);
        replay_zero_grad_steps<VEC>(k, w, m, v, l, c - l);
        row_st<VEC>(t.w + o, w);
        row_st<VEC>(t.m + o, m);
        row_st<VEC>(t.v + o, v);
    } else if (a.pay) {
        row_ld<VEC>(t.w + o, w);
    }
    if (a.pay) {
        const int* sfc = a.seg_first_chunk + (long long)f * (a.cap + 1);
        const int* cst = a.chunk_start + (long long)f * (a.cap + 1);
        const int e0 = cst[sfc[u]], e1 = cst[sfc[u + 1]];
        float* dst = a.pay + a.pay_col[ti] + cc * VEC;
        for (int e = e0; e < e1; ++e) row_st<VEC>(dst + (long long)a.sorted_src[(long long)f * a.cap + e] * a.ldp, w);
    }
    if (cc == 0) {
        if (a.mode == 2) { if (l < c) t.last[row] = c; }
        else t.last[row] = c + 1;
    }
}

// Workgroup bx of the seg_catchup_blocks() that serve field f: the first nb0 replay rows of t[0] (VEC floats a lane),
// the others rows of the 1-dim t[1] (a lane per row) -- the two tables' replay loops run in different waves.
template <int VEC, int NT>
__device__ __forceinline__ void catchup_seg_block(const SegCatchArgs& a, int f, int bx) {
    if (bx < a.nb0) catchup_seg_table<VEC, NT>(a, 0, f, bx);
    else if (a.t[1].d % 4 == 0) catchup_seg_table<4, NT>(a, 1, f, bx - a.nb0);
    else catchup_seg_table<1, NT>(a, 1, f, bx - a.nb0);
}

inline int fill_seg_catchup(SegCatchArgs& a, const int* seg_row, const int* counts, int F, int cap, int d, float* table,
                            float* m, float* v, int* last, float* table1, float* m1, float* v1, int* last1, int opt,
                            const long long* step, const float* hyper, const float* bctab, int tab_n, int kmax, int mode) {
    a.seg_row = seg_row; a.counts = counts; a.F = F; a.cap = cap;
    int nt = 0;
    if (table) {
        if (!(m && v && last) || !(d >= 1 && ((d % 4 == 0 && d / 4 <= 256) || d <= 64))) return -1;
        a.t[nt++] = RowTab{table, m, v, last, d};
    }
    if (table1) {
        if (!(m1 && v1 && last1)) return -1;
        a.t[nt++] = RowTab{table1, m1, v1, last1, 1};
    }
    a.nt = nt;
    a.opt = opt; a.step = step; a.hyper = hyper; a.bctab = bctab; a.tab_n = tab_n; a.kmax = kmax; a.mode = mode;
    return nt;
}

// fills a.t[] / scalars from entry-point arguments; returns the number of tables or -1
inline int fill_catchup(CatchArgs& a, const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, int B, int F, int d,
                        float* table, float* m, float* v, int* last, float* table1, float* m1, float* v1, int* last1, int opt,
                        const long long* step, int mark_ahead, const float* hyper, const float* bctab, int tab_n, int kmax) {
    a.idx = idx; a.row_offset = row_offset; a.vocab = vocab; a.B = B; a.F = F;
    a.sub_off = nullptr; a.sub_cnt = nullptr; a.nlists = 0; a.per_list = 0;
    int nt = 0;
    if (table) {
        if (!(m && v && last) || !(d >= 1 && ((d % 4 == 0 && d / 4 <= 256) || d <= 64))) return -1;
        a.t[nt++] = RowTab{table, m, v, last, d};
    }
    if (table1) {
        if (!(m1 && v1 && last1)) return -1;
        a.t[nt++] = RowTab{table1, m1, v1, last1, 1};
    }
    a.opt = opt; a.step = step; a.mark_ahead = mark_ahead; a.hyper = hyper; a.bctab = bctab; a.tab_n = tab_n; a.kmax = kmax;
    return nt;
}

}  // namespace

