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
// term is below 2^-30 of the weight (the remaining tail is < 10 terms of that size) or after kmax steps; what is
// left of the skipped steps only decays m, v (and, for AdamW, w) and is applied in closed form.
#include "common.h"

namespace {

enum { OPT_SGD = 0, OPT_ADAM = 1, OPT_ADAMW = 2, OPT_ADAGRAD = 3 };

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
    const bool adamw = c.opt == OPT_ADAMW;
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
        while (done < nloop) {
            ++done;
            const float2 bc = bias_corrections(c, s + done);
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
                live |= fabsf(term) > 9.3e-10f * fabsf(w[e]);
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
__device__ __forceinline__ void ldv(const float* p, float (&x)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
    } else {
        x[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&x)[VEC]) {
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
    else *p = x[0];
}

template <int VEC>
__device__ __forceinline__ void replay_row_slice(const ReplayConsts& c, const RowTab& t, long long row, int co, int s, int n) {
    const long long o = row * t.d + co;
    float w[VEC], m[VEC], v[VEC];
    ldv<VEC>(t.w + o, w);
    ldv<VEC>(t.m + o, m);
    ldv<VEC>(t.v + o, v);
    replay_zero_grad_steps<VEC>(c, w, m, v, s, n);
    stv<VEC>(t.w + o, w);
    stv<VEC>(t.m + o, m);
    stv<VEC>(t.v + o, v);
}

struct CatchArgs {
    const int64_t* idx; const int64_t* row_offset; const int64_t* vocab;
    int B, F;
    RowTab t[2];
    int opt; const long long* step; int mark_ahead;
    const float* hyper; const float* bctab; int tab_n; int kmax;
};

constexpr int CT = 256;

// Phase 1: one thread per (field, sample) entry looks at its row's stamp; a row that is behind is claimed with
// atomicMax(last[row], steps + mark_ahead): exactly one of the entries that name it sees the old value and becomes its
// owner (whichever it is, the result is the same).  Rows that are up to date -- the hot rows of a Zipf batch, stepped
// and stamped by the previous step's update -- cost one load and no atomic.
// Phase 2: the workgroup replays the rows its threads own, d/VEC lanes per row.
template <int VEC>
__device__ __forceinline__ void catchup_body(const CatchArgs& a, const RowTab& t, int* s_n, int* s_row, int* s_last) {
    if (threadIdx.x == 0) *s_n = 0;
    __syncthreads();
    const long long e = (long long)blockIdx.x * CT + threadIdx.x;
    const int c = (int)*a.step;
    if (e < (long long)a.B * a.F) {
        const int f = (int)(e / a.B);
        const long long key = a.idx[e];
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
    const int C = t.d / VEC, gpb = CT / C;
    const int grp = threadIdx.x / C, cc = threadIdx.x - grp * C;
    if (grp >= gpb) return;
    const ReplayConsts k = load_consts(a.hyper, a.opt, a.bctab, a.tab_n, a.kmax);
    for (int i = grp; i < items; i += gpb) replay_row_slice<VEC>(k, t, s_row[i], cc * VEC, s_last[i], c - s_last[i]);
}

__global__ __launch_bounds__(CT) void adam_catchup_kernel(CatchArgs a) {
    __shared__ int s_n, s_row[CT], s_last[CT];
    const RowTab t = a.t[blockIdx.y];
    if (t.d % 4 == 0) catchup_body<4>(a, t, &s_n, s_row, s_last);
    else catchup_body<1>(a, t, &s_n, s_row, s_last);
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
    RS_CHECK_ARG(opt == OPT_ADAM || opt == OPT_ADAMW, "rs_adam_catchup: optimizer %d has nothing to replay", opt);
    RS_CHECK_ARG(B > 0 && F > 0 && kmax >= 0 && tab_n >= 0 && (tab_n == 0 || bctab), "rs_adam_catchup: bad sizes");
    RS_CHECK_ARG(mark_ahead == 0 || mark_ahead == 1, "rs_adam_catchup: mark_ahead must be 0 or 1");
    CatchArgs a{};
    a.idx = idx; a.row_offset = row_offset; a.vocab = vocab; a.B = B; a.F = F;
    int nt = 0;
    if (table) {
        RS_CHECK_ARG(m && v && last && row_width_ok(d), "rs_adam_catchup: table needs m, v, last and a row width the kernels take (%d)", d);
        a.t[nt++] = RowTab{table, m, v, last, d};
    }
    if (table1) {
        RS_CHECK_ARG(m1 && v1 && last1, "rs_adam_catchup: table1 needs m1, v1, last1");
        a.t[nt++] = RowTab{table1, m1, v1, last1, 1};
    }
    RS_CHECK_ARG(nt > 0, "rs_adam_catchup: no table");
    a.opt = opt; a.step = step; a.mark_ahead = mark_ahead; a.hyper = hyper; a.bctab = bctab; a.tab_n = tab_n; a.kmax = kmax;
    hipLaunchKernelGGL(adam_catchup_kernel, dim3(rs_cdiv((long long)B * F, CT), nt), dim3(CT), 0, stream, a);
    RS_CHECK_LAUNCH("rs_adam_catchup");
    return RS_OK;
}

int rs_adam_flush(float* table, float* m, float* v, int* last, long long rows, int d, int opt, const long long* step,
                  const float* hyper, const float* bctab, int tab_n, int kmax, hipStream_t stream) {
    RS_CHECK_ARG(table && m && v && last && step && hyper, "rs_adam_flush: null pointer");
    RS_CHECK_ARG(opt == OPT_ADAM || opt == OPT_ADAMW, "rs_adam_flush: optimizer %d has nothing to replay", opt);
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
