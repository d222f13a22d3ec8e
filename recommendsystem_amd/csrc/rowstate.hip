// Dense-Adam semantics on tables that are stepped only where a batch touches them.
//
// The reference steps every row of every table with torch.optim.Adam / AdamW on each batch
// (model/utils/BaseModel.py:64-72,149; the tables are dense nn.Embedding parameters, model/utils/Layers.py:132).
// A row that a batch does not touch receives a zero gradient there: its moments decay
// (m <- m + (0 - m)(1 - beta1), v <- v beta2), AdamW shrinks it by (1 - lr wd), and it still moves by
// lr/(1 - beta1^k) * m / (sqrt(v)/sqrt(1 - beta2^k) + eps) while m is not zero.  The kernels here replay those
// zero-gradient steps for a row when it is next read ("catch-up on touch"): every table carries one int per row,
// last[row] = the number of optimizer steps already contained in (w, m, v) of that row.
//
//   rs_adam_catchup   before a lookup: brings the rows a batch is about to read up to the current step count
//   rs_adam_flush     brings every row of a table up to the current step count (evaluation, checkpoints, a change
//                     of the learning rate)
//   rs_adam_bc_table  the per-step scalars 1/(1 - beta1^k) and 1/sqrt(1 - beta2^k), computed in double precision
//                     as torch computes them on the host
//
// The replay is exact up to rounding: the drift terms decay like (beta1/sqrt(beta2))^j, so the loop stops once a
// term is below 2^-25 of the weight (under half an ulp: neither it nor any later, smaller term changes the weight in
// the dense optimizer's fp32 arithmetic either; looked at every eighth step) or
// after kmax steps; what is left of the skipped steps only decays m, v (and, for AdamW, w) and is applied in closed
// form.
#include "rowstate_body.h"
#include "plan.h"

namespace {

constexpr int CT = 256;

__global__ __launch_bounds__(CT) void adam_catchup_kernel(CatchArgs a, int per_table) {
    catchup_block<CT>(a, blockIdx.x, per_table);
}

// the distinct rows of a sorted batch (rowstate_body.h, catchup_seg_block); grid (seg_catchup_blocks(), F)
template <int VEC>
__global__ __launch_bounds__(CT) void adam_catchup_plan_kernel(SegCatchArgs a) {
    catchup_seg_block<VEC, CT>(a, blockIdx.y, blockIdx.x);
}

// The same for rows of more than 256 floats (field-aware FFM rows, F*d up to 1024): a wave per distinct row, NS float4
// slices per lane, the waves striding over the F lists of distinct rows laid end to end (F <= 64: lane f holds field f's
// count).  A row whose moments are all zero does not move under Adam and is left unread; rows of the 1-dim second table
// are served by the workgroups behind the first wide_blocks, a lane per row.
template <int NS>
__global__ __launch_bounds__(CT) void adam_catchup_plan_wide_kernel(SegCatchArgs a, int wide_blocks, int nb1) {
    if ((int)blockIdx.x >= wide_blocks) {
        const int b = blockIdx.x - wide_blocks;
        catchup_seg_table<1, CT>(a, 1, b / nb1, b % nb1);
        return;
    }
    const RowTab t = a.t[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (int)*a.step;
    const int n = lane < a.F ? a.counts[lane * 4 + 2] : 0;
    int end = n;                                              // inclusive prefix over the fields
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(end, o);
        if (lane >= o) end += y;
    }
    const int total = __shfl(end, 63);
    const int C4 = t.d / 4;
    const ReplayConsts k = load_consts(a.hyper, a.opt, a.bctab, a.tab_n, a.kmax);
    for (int it = blockIdx.x * (CT / 64) + wave; it < total; it += wide_blocks * (CT / 64)) {
        const int f = __popcll(__ballot(end <= it));          // fields that lie wholly in front of item `it`
        const int u = it - (__shfl(end, f) - __shfl(n, f));
        const int row = a.seg_row[(long long)f * a.cap + u];
        const int l = t.last[row];
        if (a.mode == 1) {
            if (lane == 0 && l > c) t.last[row] = c;
            continue;
        }
        if (l < c) {
            constexpr int V = 4 * NS;
            float w[V], m[V], v[V];
            float* const pw = t.w + (long long)row * t.d;
            float* const pm = t.m + (long long)row * t.d;
            float* const pv = t.v + (long long)row * t.d;
            bool nz = false;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int i = lane + 64 * s;
                float4 mm = make_float4(0.f, 0.f, 0.f, 0.f), vv = mm;
                if (i < C4) {
                    mm = *reinterpret_cast<const float4*>(pm + 4 * i);
                    vv = *reinterpret_cast<const float4*>(pv + 4 * i);
                }
                m[4 * s] = mm.x; m[4 * s + 1] = mm.y; m[4 * s + 2] = mm.z; m[4 * s + 3] = mm.w;
                v[4 * s] = vv.x; v[4 * s + 1] = vv.y; v[4 * s + 2] = vv.z; v[4 * s + 3] = vv.w;
                nz |= (mm.x != 0.f) | (mm.y != 0.f) | (mm.z != 0.f) | (mm.w != 0.f) | (vv.x != 0.f) | (vv.y != 0.f) |
                      (vv.z != 0.f) | (vv.w != 0.f);
            }
            if (a.opt == RS_OPT_ADAMW || __ballot(nz) != 0ull) {              // uniform over the wave
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int i = lane + 64 * s;
                    float4 ww = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i < C4) ww = *reinterpret_cast<const float4*>(pw + 4 * i);
                    w[4 * s] = ww.x; w[4 * s + 1] = ww.y; w[4 * s + 2] = ww.z; w[4 * s + 3] = ww.w;
                }
                replay_zero_grad_steps<V>(k, w, m, v, l, c - l);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int i = lane + 64 * s;
                    if (i < C4) {
                        *reinterpret_cast<float4*>(pw + 4 * i) = make_float4(w[4 * s], w[4 * s + 1], w[4 * s + 2], w[4 * s + 3]);
                        *reinterpret_cast<float4*>(pm + 4 * i) = make_float4(m[4 * s], m[4 * s + 1], m[4 * s + 2], m[4 * s + 3]);
                        *reinterpret_cast<float4*>(pv + 4 * i) = make_float4(v[4 * s], v[4 * s + 1], v[4 * s + 2], v[4 * s + 3]);
                    }
                }
            }
        }
        if (lane == 0) {
            if (a.mode == 2) { if (l < c) t.last[row] = c; }
            else t.last[row] = c + 1;
        }
    }
}

// every row of one table up to the current step count
template <int VEC>
__global__ __launch_bounds__(CT) void adam_flush_kernel(RowTab t, long long rows, int opt, const long long* step,
                                                        const float* hyper, const float* bctab, int tab_n, int kmax) {
    const int C = t.d / VEC, gpb = CT / C;
    const int grp = threadIdx.x / C, cc = threadIdx.x - grp * C;
    const long long row = (long long)blockIdx.x * gpb + grp;
    const bool active = grp < gpb && row < rows;
    const int c = (int)*step;
    const int s = active ? t.last[row] : c;
    __syncthreads();                                    // every lane of a row has read last[row] before it changes
    if (!active || s >= c) return;
    const ReplayConsts k = load_consts(hyper, opt, bctab, tab_n, kmax);
    replay_row_slice<VEC>(k, t, row, cc * VEC, s, c - s);
    if (cc == 0) t.last[row] = c;
}

__global__ void adam_bc_table_kernel(float2* tab, int n, const float* __restrict__ hyper) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (k == 0) { tab[0] = make_float2(0.f, 0.f); return; }
    const double b1 = (double)hyper[1], b2 = (double)hyper[2];
    tab[k] = make_float2((float)(1.0 / (1.0 - pow(b1, (double)k))), (float)(1.0 / sqrt(1.0 - pow(b2, (double)k))));
}

bool row_width_ok(int d) { return d >= 1 && ((d % 4 == 0 && d / 4 <= CT) || d <= 64); }

}  // namespace

extern "C" {

int rs_adam_bc_table(float* tab, int n, const float* hyper, hipStream_t stream) {
    RS_CHECK_ARG(tab && hyper && n > 0, "rs_adam_bc_table: null pointer or empty table");
    hipLaunchKernelGGL(adam_bc_table_kernel, dim3(rs_cdiv(n, 256)), dim3(256), 0, stream, reinterpret_cast<float2*>(tab), n,
                       hyper);
    RS_CHECK_LAUNCH("rs_adam_bc_table");
    return RS_OK;
}

int rs_adam_catchup(const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, int B, int F, int d,
                    float* table, float* m, float* v, int* last, float* table1, float* m1, float* v1, int* last1,
                    int opt, const long long* step, int mark_ahead, const float* hyper, const float* bctab, int tab_n,
                    int kmax, hipStream_t stream) {
    RS_CHECK_ARG(idx && row_offset && vocab && step && hyper, "rs_adam_catchup: null pointer");
    RS_CHECK_ARG(opt == RS_OPT_ADAM || opt == RS_OPT_ADAMW, "rs_adam_catchup: optimizer %d has nothing to replay", opt);
    RS_CHECK_ARG(B > 0 && F > 0 && kmax >= 0 && tab_n >= 0 && (tab_n == 0 || bctab), "rs_adam_catchup: bad sizes");
    RS_CHECK_ARG(mark_ahead == 0 || mark_ahead == 1, "rs_adam_catchup: mark_ahead must be 0 or 1");
    CatchArgs a{};
    const int nt = fill_catchup(a, idx, row_offset, vocab, B, F, d, table, m, v, last, table1, m1, v1, last1, opt, step,
                                mark_ahead, hyper, bctab, tab_n, kmax);
    RS_CHECK_ARG(nt > 0, "rs_adam_catchup: no table, or a table without m / v / last, or a row width the kernels do not take (%d)", d);
    const int per_table = rs_cdiv((long long)B * F, CT);
    hipLaunchKernelGGL(adam_catchup_kernel, dim3(per_table * nt), dim3(CT), 0, stream, a, per_table);
    RS_CHECK_LAUNCH("rs_adam_catchup");
    return RS_OK;
}

int rs_adam_catchup_lists(const int64_t* keys, const int* sub_off, const int* sub_cnt, int nlists, int per_list,
                          const int64_t* row_offset, const int64_t* vocab, int F, int d, float* table, float* m, float* v,
                          int* last, float* table1, float* m1, float* v1, int* last1, int opt, const long long* step,
                          int mark_ahead, const float* hyper, const float* bctab, int tab_n, int kmax, hipStream_t stream) {
    RS_CHECK_ARG(keys && sub_off && sub_cnt && row_offset && vocab && step && hyper, "rs_adam_catchup_lists: null pointer");
    RS_CHECK_ARG(opt == RS_OPT_ADAM || opt == RS_OPT_ADAMW, "rs_adam_catchup_lists: optimizer %d has nothing to replay", opt);
    RS_CHECK_ARG(F > 0 && nlists > 0 && nlists % F == 0 && per_list > 0, "rs_adam_catchup_lists: %d lists of up to %d keys for %d fields",
                 nlists, per_list, F);
    RS_CHECK_ARG(kmax >= 0 && tab_n >= 0 && (tab_n == 0 || bctab), "rs_adam_catchup_lists: bad sizes");
    RS_CHECK_ARG(mark_ahead == 0 || mark_ahead == 1, "rs_adam_catchup_lists: mark_ahead must be 0 or 1");
    CatchArgs a{};
    const int nt = fill_catchup(a, keys, row_offset, vocab, 0, F, d, table, m, v, last, table1, m1, v1, last1, opt, step,
                                mark_ahead, hyper, bctab, tab_n, kmax);
    RS_CHECK_ARG(nt > 0, "rs_adam_catchup_lists: no table, or a table without m / v / last, or a row width the kernels do not take (%d)", d);
    a.sub_off = sub_off; a.sub_cnt = sub_cnt; a.nlists = nlists; a.per_list = per_list;
    const int per_table = rs_cdiv((long long)nlists * per_list, CT);
    hipLaunchKernelGGL(adam_catchup_kernel, dim3(per_table * nt), dim3(CT), 0, stream, a, per_table);
    RS_CHECK_LAUNCH("rs_adam_catchup_lists");
    return RS_OK;
}

// The replay for the distinct rows of a sorted batch (`plan`: rs_sparse_bwd_sort's, F fields of capacity cap), with the
// rows promised to the next update: mode 0 brings every row of the plan whose stamp is below *step up to *step and
// stamps all of them *step + 1 (the update that follows must then be given no stamps); mode 1 takes the promise back
// (stamps above *step return to *step) when that update does not come; mode 2 is rs_adam_catchup per distinct row
// (rows below *step brought up to it and stamped *step; the update that follows stamps as usual).  No atomics: a plan
// names a row once.
static int catchup_plan_impl(const int* plan, int F, int cap, int d, float* table, float* m, float* v, int* last, float* table1,
                             float* m1, float* v1, int* last1, int opt, const long long* step, const float* hyper,
                             const float* bctab, int tab_n, int kmax, int mode, float* pay, int ld_pay, int col, int col1,
                             hipStream_t stream) {
    RS_CHECK_ARG(plan && step && hyper, "rs_adam_catchup_plan: null pointer");
    RS_CHECK_ARG(opt == RS_OPT_ADAM || opt == RS_OPT_ADAMW, "rs_adam_catchup_plan: optimizer %d has nothing to replay", opt);
    RS_CHECK_ARG(F > 0 && cap > 0 && kmax >= 0 && tab_n >= 0 && (tab_n == 0 || bctab), "rs_adam_catchup_plan: bad sizes");
    RS_CHECK_ARG(mode >= 0 && mode <= 2, "rs_adam_catchup_plan: mode must be 0, 1 or 2");
    const PlanView pv = carve_plan(const_cast<int*>(plan), F, cap);
    SegCatchArgs a{};
    const int nt = fill_seg_catchup(a, pv.seg_row, pv.counts, F, cap, d, table, m, v, last, table1, m1, v1, last1, opt, step,
                                    hyper, bctab, tab_n, kmax, mode);
    RS_CHECK_ARG(nt > 0, "rs_adam_catchup_plan: no table, or a table without m / v / last, or a row width the kernels do not take (%d)", d);
    if (pay) {
        const int c0 = table ? col : col1;           // column of t[0]: the d-wide table when there is one
        RS_CHECK_ARG(mode == 2 && a.t[0].d <= 256, "rs_adam_catchup_serve: rows are served by the plain replay of rows up to 256 floats");
        RS_CHECK_ARG(ld_pay >= 1 && c0 >= 0 && c0 + a.t[0].d <= ld_pay && (nt < 2 || (col1 >= 0 && col1 < ld_pay)),
                     "rs_adam_catchup_serve: columns %d / %d do not fit positions of %d floats", col, col1, ld_pay);
        RS_CHECK_ARG(a.t[0].d % 4 != 0 || (ld_pay % 4 == 0 && c0 % 4 == 0 && (reinterpret_cast<uintptr_t>(pay) & 15) == 0),
                     "rs_adam_catchup_serve: rows of %d floats need 16-byte aligned positions", a.t[0].d);
        a.pay = pay; a.ldp = ld_pay; a.pay_col[0] = c0; a.pay_col[1] = col1;
        a.sorted_src = pv.sorted_src; a.chunk_start = pv.chunk_start; a.seg_first_chunk = pv.seg_first_chunk;
    }
    if (a.t[0].d > 256 && a.t[0].d % 4 == 0 && F <= 64) {
        // wide rows: a wave per row instead of a workgroup per plan slot
        seg_catchup_blocks(a, CT);
        const int nb1 = nt > 1 ? rs_cdiv(cap, seg_rows_per_block(a.t[1].d, CT)) : 0;
        const long long slots = (long long)F * cap;
        const int wide_blocks = (int)(slots / (CT / 64) + 1 < 2048 ? slots / (CT / 64) + 1 : 2048);
        const dim3 wgrid(wide_blocks + F * nb1);
        const int ns = rs_cdiv(a.t[0].d / 4, 64);
        if (ns == 1) hipLaunchKernelGGL(adam_catchup_plan_wide_kernel<1>, wgrid, dim3(CT), 0, stream, a, wide_blocks, nb1);
        else if (ns == 2) hipLaunchKernelGGL(adam_catchup_plan_wide_kernel<2>, wgrid, dim3(CT), 0, stream, a, wide_blocks, nb1);
        else if (ns == 3) hipLaunchKernelGGL(adam_catchup_plan_wide_kernel<3>, wgrid, dim3(CT), 0, stream, a, wide_blocks, nb1);
        else hipLaunchKernelGGL(adam_catchup_plan_wide_kernel<4>, wgrid, dim3(CT), 0, stream, a, wide_blocks, nb1);
        RS_CHECK_LAUNCH("rs_adam_catchup_plan");
        return RS_OK;
    }
    const dim3 grid(seg_catchup_blocks(a, CT), F);
    if (a.t[0].d % 4 == 0) hipLaunchKernelGGL(adam_catchup_plan_kernel<4>, grid, dim3(CT), 0, stream, a);
    else hipLaunchKernelGGL(adam_catchup_plan_kernel<1>, grid, dim3(CT), 0, stream, a);
    RS_CHECK_LAUNCH("rs_adam_catchup_plan");
    return RS_OK;
}

int rs_adam_catchup_plan(const int* plan, int F, int cap, int d, float* table, float* m, float* v, int* last, float* table1,
                         float* m1, float* v1, int* last1, int opt, const long long* step, const float* hyper,
                         const float* bctab, int tab_n, int kmax, int mode, hipStream_t stream) {
    return catchup_plan_impl(plan, F, cap, d, table, m, v, last, table1, m1, v1, last1, opt, step, hyper, bctab, tab_n, kmax,
                             mode, nullptr, 0, 0, 0, stream);
}

// rs_adam_catchup_plan(mode 2) that also SERVES the rows: every row of the plan, once up to date, is written to each
// position that names it (the plan's entries of its segment: sorted_src) of `pay`, a buffer of ld_pay floats per
// position -- the d-wide row at column `col`, the 1-dim value at column `col1`.  What rs_shard_gather does for the owner
// of a row-sharded table, without a second pass over the rows the replay has just touched.  Rows of up to 256 floats.
int rs_adam_catchup_serve(const int* plan, int F, int cap, int d, float* table, float* m, float* v, int* last, float* table1,
                          float* m1, float* v1, int* last1, int opt, const long long* step, const float* hyper,
                          const float* bctab, int tab_n, int kmax, float* pay, int ld_pay, int col, int col1,
                          hipStream_t stream) {
    RS_CHECK_ARG(pay, "rs_adam_catchup_serve: null row buffer");
    return catchup_plan_impl(plan, F, cap, d, table, m, v, last, table1, m1, v1, last1, opt, step, hyper, bctab, tab_n, kmax,
                             2, pay, ld_pay, col, col1, stream);
}

int rs_adam_flush(float* table, float* m, float* v, int* last, long long rows, int d, int opt, const long long* step,
                  const float* hyper, const float* bctab, int tab_n, int kmax, hipStream_t stream) {
    RS_CHECK_ARG(table && m && v && last && step && hyper, "rs_adam_flush: null pointer");
    RS_CHECK_ARG(opt == RS_OPT_ADAM || opt == RS_OPT_ADAMW, "rs_adam_flush: optimizer %d has nothing to replay", opt);
    RS_CHECK_ARG(rows > 0 && row_width_ok(d) && kmax >= 0 && tab_n >= 0 && (tab_n == 0 || bctab), "rs_adam_flush: bad sizes");
    const RowTab t{table, m, v, last, d};
    if (d % 4 == 0) {
        const int gpb = CT / (d / 4);
        hipLaunchKernelGGL(adam_flush_kernel<4>, dim3(rs_cdiv(rows, gpb)), dim3(CT), 0, stream, t, rows, opt, step, hyper,
                           bctab, tab_n, kmax);
    } else {
        const int gpb = CT / d;
        hipLaunchKernelGGL(adam_flush_kernel<1>, dim3(rs_cdiv(rows, gpb)), dim3(CT), 0, stream, t, rows, opt, step, hyper,
                           bctab, tab_n, kmax);
    }
    RS_CHECK_LAUNCH("rs_adam_flush");
    return RS_OK;
}

}  // extern "C"
