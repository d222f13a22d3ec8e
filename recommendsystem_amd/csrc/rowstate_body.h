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
    bool tail_one;      // the table's last entry is {1, 1} in float: so is every later step's (both corrections fall to 1)
};

__device__ __forceinline__ ReplayConsts load_consts(const float* __restrict__ hyper, int opt, const float* bctab, int tab_n,
                                                    int kmax) {
    ReplayConsts c;
    c.lr = hyper[0]; c.b1 = hyper[1]; c.b2 = hyper[2]; c.eps = hyper[3]; c.wd = hyper[4];
    c.opt = opt; c.kmax = kmax; c.tab_n = tab_n;
    c.bctab = reinterpret_cast<const float2*>(bctab);
    c.tail_one = false;
    if (tab_n > 1) {
        const float2 t = c.bctab[tab_n - 1];
        c.tail_one = t.x == 1.f && t.y == 1.f;
    }
    return c;
}

// {1/(1 - beta1^k), 1/sqrt(1 - beta2^k)}
__device__ __forceinline__ float2 bias_corrections(const ReplayConsts& c, int k) {
    if (k < c.tab_n) return c.bctab[k];
    if (c.tail_one) return make_float2(1.f, 1.f);
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

// b of any of the four lanes 4q .. 4q+3 this lane belongs to (all four must be executing)
__device__ __forceinline__ bool quad_any(bool b) {
    int x = b ? 1 : 0;
    x |= __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]
    x |= __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
    return x != 0;
}

// (w, m, v) after the zero-gradient optimizer steps s+1 .. s+n.  The loop runs while any of the lane's VEC elements
// still moves its weight.  QUAD (VEC = 1): four neighbouring lanes hold the four elements that one lane of the VEC = 4
// form holds and leave the loop together, so both forms execute the same arithmetic on every element -- the same bits --
// and a caller may pick either one by how many lanes it has to spare.
template <int VEC, bool QUAD = false>
__device__ __forceinline__ void replay_zero_grad_steps(const ReplayConsts& c, float (&w)[VEC], float (&m)[VEC],
                                                       float (&v)[VEC], int s, int n) {
    static_assert(!QUAD || VEC == 1, "the quad form holds one element per lane");
    const bool adamw = c.opt == RS_OPT_ADAMW;
    const float omb1 = 1.f - c.b1;
    const float decay = 1.f - c.lr * c.wd;
    bool moving = false;
#pragma unroll
    for (int e = 0; e < VEC; ++e) moving |= (m[e] != 0.f);
    if constexpr (QUAD) moving = quad_any(moving);
    int done = 0;
    if (moving) {
        float sq[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) sq[e] = sqrtf(v[e]);
        const float rb2 = sqrtf(c.b2);
        const int nloop = n < c.kmax ? n : c.kmax;
        const int klast = s + nloop;
        // one zero-gradient step with its scalars bc = {1/(1 - beta1^k), 1/sqrt(1 - beta2^k)}; false once no element
        // of this lane moves its weight any more
        auto one_step = [&](const float2 bc) -> bool {
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
                // (below 2^-25 |w| -- less than half an ulp of w -- the subtraction leaves w as it is, in the dense
                //  optimizer's fp32 arithmetic as well, and the terms only get smaller from step to step)
                live |= fabsf(term) > 2.9e-8f * fabsf(w[e]);
            }
            return live;
        };
        if (klast < c.tab_n || c.tail_one) {
            // The scalars come from the table, one entry per step.  One step's arithmetic is far shorter than the latency
            // of that load and a long replay is one dependent chain, so the steps go in groups of NQ with NQ entries in
            // flight, each in a register pair of its own that is refilled right after its step has used it, and no
            // branch inside a group (a load under a condition, or a rotating queue, makes every step wait for the
            // newest load).  Whether the weights still move is looked at once per group: up to NQ - 1 more (real) steps
            // than needed are replayed.
            constexpr int NQ = 8;
            float2 q[NQ];
            const int kcap = min(klast, c.tab_n - 1);      // (steps past the table: its last entry, {1, 1})
#pragma unroll
            for (int j = 0; j < NQ; ++j) q[j] = c.bctab[min(s + 1 + j, kcap)];
            bool live = true;
            while (done + NQ <= nloop) {
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    live = one_step(q[j]);
                    q[j] = c.bctab[min(s + done + 1 + NQ + j, kcap)];
                }
                done += NQ;
                if constexpr (QUAD) live = quad_any(live);
                if (!live) break;
            }
            if (live) {                                    // fewer than NQ steps are left: q[j] holds step done+1+j
                const int r = nloop - done;
#pragma unroll
                for (int j = 0; j < NQ - 1; ++j)
                    if (j < r) one_step(q[j]);
                done = nloop;
            }
        } else {
            float2 bc_next = bias_corrections(c, s + 1);
            while (done < nloop) {
                ++done;
                const float2 bc = bc_next;
                bc_next = bias_corrections(c, s + done + 1);
                bool live = one_step(bc);
                if constexpr (QUAD) live = quad_any(live);
                if (!live) break;
            }
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

template <int VEC, bool QUAD = false>
__device__ __forceinline__ void replay_row_slice(const ReplayConsts& c, const RowTab& t, long long row, int co, int s, int n) {
    const long long o = row * t.d + co;
    float w[VEC], m[VEC], v[VEC];
    row_ld<VEC>(t.w + o, w);
    row_ld<VEC>(t.m + o, m);
    row_ld<VEC>(t.v + o, v);
    replay_zero_grad_steps<VEC, QUAD>(c, w, m, v, s, n);
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
// Rows of up to FINE_MAX_D floats (a multiple of four) whose first moment is not zero -- the ones whose replay is a loop
// of up to kmax steps -- are moved to the front of the list first; when they are few enough to leave lanes idle, a lane
// per element replays them (the quad form above) instead of a lane per four elements: a quarter of the time per step.
// NT threads per workgroup; `block` = index of the workgroup among those that do this work; s_row, s_last, s_row2,
// s_last2: NT ints each; s_n: 3 ints.
constexpr int FINE_MAX_D = 64;
template <int VEC, int NT>
__device__ __forceinline__ void catchup_body(const CatchArgs& a, const RowTab& t, int block, int* s_n, int* s_row, int* s_last,
                                             int* s_row2, int* s_last2) {
    if (threadIdx.x < 3) s_n[threadIdx.x] = 0;
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
    const ReplayConsts k = load_consts(a.hyper, a.opt, a.bctab, a.tab_n, a.kmax);
    if constexpr (VEC == 4) {
        if (t.d <= FINE_MAX_D && NT % t.d == 0) {                          // uniform over the workgroup
            // the rows with a loop to run in front (s_row2[0, nh)), the others from the back
            if ((int)threadIdx.x < items) {
                const int row = s_row[threadIdx.x];
                bool mv = false;
                for (int q = 0; q < C; ++q) {
                    const float4 mm = *reinterpret_cast<const float4*>(t.m + (long long)row * t.d + 4 * q);
                    mv |= (mm.x != 0.f) | (mm.y != 0.f) | (mm.z != 0.f) | (mm.w != 0.f);
                }
                const int p = mv ? atomicAdd(s_n + 1, 1) : NT - 1 - atomicAdd(s_n + 2, 1);
                s_row2[p] = row; s_last2[p] = s_last[threadIdx.x];
            }
            __syncthreads();
            const int nh = s_n[1], nl = s_n[2];
            const int gpb1 = NT / t.d;
            if (nh <= 3 * gpb1) {
                const int grp1 = threadIdx.x / t.d, cc1 = threadIdx.x - grp1 * t.d;
                for (int i = grp1; i < nh; i += gpb1)
                    replay_row_slice<1, true>(k, t, s_row2[i], cc1, s_last2[i], c - s_last2[i]);
            } else if (grp < gpb) {
                for (int i = grp; i < nh; i += gpb) replay_row_slice<4>(k, t, s_row2[i], cc * 4, s_last2[i], c - s_last2[i]);
            }
            if (grp < gpb)
                for (int i = grp; i < nl; i += gpb)
                    replay_row_slice<4>(k, t, s_row2[NT - 1 - i], cc * 4, s_last2[NT - 1 - i], c - s_last2[NT - 1 - i]);
            return;
        }
    }
    if (grp >= gpb) return;
    for (int i = grp; i < items; i += gpb) replay_row_slice<VEC>(k, t, s_row[i], cc * VEC, s_last[i], c - s_last[i]);
}

// workgroup `block` of the nt * ceil(B*F / NT) that serve a batch: table block / per_table, entries of block % per_table
template <int NT>
__device__ __forceinline__ void catchup_block(const CatchArgs& a, int block, int per_table) {
    __shared__ int s_n[3], s_row[NT], s_last[NT], s_row2[NT], s_last2[NT];
    const RowTab t = a.t[block / per_table];
    if (t.d % 4 == 0) catchup_body<4, NT>(a, t, block % per_table, s_n, s_row, s_last, s_row2, s_last2);
    else catchup_body<1, NT>(a, t, block % per_table, s_n, s_row, s_last, s_row2, s_last2);
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

