// Row-sharded embedding tables: the device side of the all-to-all exchange (one process per GPU).
//
// Row i of every field lives on rank i % N at local index i / N + 1 (local row 0 of every field is a
// zero dummy, so "index 0 = padding, never trained" holds on every rank).  A rank first sorts and
// de-duplicates its own batch per field (field_sort_kernel), so only DISTINCT rows travel:
//
//   requester  route_kernel        distinct keys of field f -> per-owner buckets of send_idx (slot 0 = count, then the
//                                  keys), route[f][u] = position of the key's row in the row / gradient buffers
//   (all-to-all of send_idx)
//   owner      shard_gather_kernel rows for every received local index -> rows[s][f][j][:], rows1[s][f][j]
//   (all-to-all of rows, rows1)
//   requester  shard_lookup_kernel the fused gather of the single-GPU path (gather_body.h) over the received rows:
//                                  entry (f, b) reads the row of its distinct key through slot_map[f][b] (written by
//                                  route_kernel) -> emb[B,F,d], sum over fields, FM term, LR sum, as K1 leaves them
//              shard_expand_kernel the plain copy for rows wider than the fused gather takes (field-aware FFM rows)
//   ... forward / backward of the dense part, local segmented reduce to one gradient per distinct key ...
//   requester  shard_pack_kernel   distinct-key gradients into the same bucket positions
//   (all-to-all of grads, grads1)
//   owner      field_sort_kernel over the N received sub-lists of each field + reduce + update (sparse.hip)
//
// Buckets have fixed capacities so that every collective is rectangular and no count ever visits the host: bucket
// (owner o, field f) holds up to capf[f] keys -- a field with few rows never needs more than its rows per owner, which
// is what keeps the all-to-alls small -- and a bucket overflow sets *overflow_flag (checked by the caller outside the
// step).  With boff[f] = capf[0] + .. + capf[f-1] and S = boff[F]:
//   key buffers   [N, S + F] int64: bucket (o, f) starts at o*(S+F) + boff[f] + f (count slot first)
//   row buffers   [N, S, ld]:       row r of bucket (o, f) is at position o*S + boff[f] + r; ld >= d floats per position,
//                                   so that the value of the 1-dim (LR) table can travel in a spare column of the same
//                                   buffer (one all-to-all each way instead of two); position N*S is a row of zeros
//                                   that padding entries and keys that did not fit their bucket are pointed at
#include "common.h"
#include "plan.h"
#include "gather_body.h"

namespace {

constexpr int RT_THREADS = 1024;

// exclusive scan of one int per thread over a 1024-thread workgroup (same as the sort's)
__device__ __forceinline__ int excl_scan_1024(int v, int* scratch, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) scratch[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < RT_THREADS / 64; ++i) {
        const int s = scratch[i];
        if (i < w) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

// One workgroup per field.  Distinct keys arrive sorted ascending in seg_row; thread t owns keys
// t*IT .. t*IT+IT-1 so that ranks inside an owner bucket keep ascending key order.
template <int IT>
__global__ __launch_bounds__(RT_THREADS) void route_kernel(const int* __restrict__ plan_seg_row,
                                                           const int* __restrict__ counts,
                                                           const int64_t* __restrict__ row_offset, int F, int cap,
                                                           int N, const int* __restrict__ capf,
                                                           const int* __restrict__ boff, int S,
                                                           long long* __restrict__ send_idx,
                                                           int* __restrict__ route, int* __restrict__ overflow_flag) {
    __shared__ int scratch[32];
    const int f = blockIdx.x, t = threadIdx.x;
    const int cap_b = capf[f], bo = boff[f];
    const int n_seg = counts[f * 4 + 2];
    const int roff = (int)row_offset[f];
    int key[IT], own[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int u = t * IT + j;
        key[j] = -1; own[j] = -1;
        if (u < n_seg) {
            key[j] = plan_seg_row[(long long)f * cap + u] - roff;
            own[j] = key[j] % N;
        }
    }
    for (int o = 0; o < N; ++o) {
        int c = 0;
#pragma unroll
        for (int j = 0; j < IT; ++j) c += (own[j] == o) ? 1 : 0;
        int total;
        int r = excl_scan_1024(c, scratch, &total);
        long long* bucket = send_idx + (long long)o * (S + F) + bo + f;
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            if (own[j] == o) {
                if (r < cap_b) {
                    bucket[1 + r] = (long long)(key[j] / N + 1);
                    route[(long long)f * cap + t * IT + j] = o * S + bo + r;
                } else {
                    route[(long long)f * cap + t * IT + j] = -1;
                }
                ++r;
            }
        }
        if (t == 0) {
            bucket[0] = total < cap_b ? total : cap_b;
            if (total > cap_b && overflow_flag) atomicOr(overflow_flag, 1);
        }
    }
}

// Requester: every entry (f, b) of the sorted batch -> the position of its key's row among the received rows (the zero
// row at zero_slot for padding entries and keys that did not fit their bucket).  Entry i of field f in key order.
__device__ __forceinline__ void slot_map_entry(const int* __restrict__ sorted_src, const int* __restrict__ entry_seg,
                                               const int* __restrict__ counts, const int* __restrict__ route, int cap,
                                               int zero_slot, int* __restrict__ slot_map, int B, int Fdiv, int f, int i) {
    if (i >= counts[f * 4 + 0]) return;
    const int u = entry_seg[(long long)f * cap + i];
    int pos = u >= 0 ? route[(long long)f * cap + u] : -1;
    if (pos < 0) pos = zero_slot;
    slot_map[(long long)f * B + sorted_src[(long long)f * cap + i] / Fdiv] = pos;
}

// Owner side: rows for the received local indices.  grid (ceil(max capf / groups_per_block), N*F).
template <int VEC>
__global__ __launch_bounds__(256) void shard_gather_kernel(const long long* __restrict__ recv_idx,
                                                           const int64_t* __restrict__ local_row_offset,
                                                           const int64_t* __restrict__ local_vocab, int F,
                                                           const int* __restrict__ capf, const int* __restrict__ boff,
                                                           int S, int d, int C, const float* __restrict__ table,
                                                           const float* __restrict__ table1, float* __restrict__ rows,
                                                           float* __restrict__ rows1, int ld, int ld1,
                                                           int* __restrict__ err_flag) {
    const int sf = blockIdx.y;                 // s*F + f
    const int src = sf / F, f = sf - src * F;
    const int grp = threadIdx.x / C, cc = threadIdx.x - grp * C;
    const int gpb = 256 / C;
    const int j = blockIdx.x * gpb + grp;
    const int cap_b = capf[f], bo = boff[f];
    if (grp >= gpb || j >= cap_b) return;
    const long long* bucket = recv_idx + (long long)src * (S + F) + bo + f;
    const int cnt = (int)bucket[0];
    if (j >= cnt) return;
    long long li = bucket[1 + j];
    if (li < 0 || li >= local_vocab[f]) { li = 0; if (err_flag) atomicOr(err_flag, 1); }
    const long long row = local_row_offset[f] + li;
    const long long o = (long long)src * S + bo + j;
    if (table) {
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(rows + o * ld + cc * 4) = *reinterpret_cast<const float4*>(table + row * d + cc * 4);
        } else {
            rows[o * ld + cc] = table[row * d + cc];
        }
    }
    if (cc == 0 && table1) rows1[o * ld1] = table1[row];
}

// Requester side, fused: K1 (gather_body.h) over the received rows, one slot per entry.
template <int VEC>
__global__ __launch_bounds__(256) void shard_lookup_kernel(GatherArgs a) {
    gather_fused_body<VEC, true>(a, (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6));
}

// Requester side: emb[src, :] = rows[route[f][entry_seg]] for every sorted entry; padding entries get zeros.
template <int VEC>
__global__ __launch_bounds__(256) void shard_expand_kernel(const int* __restrict__ sorted_src,
                                                           const int* __restrict__ entry_seg,
                                                           const int* __restrict__ counts,
                                                           const int* __restrict__ route, int F, int cap,
                                                           int d, int C, const float* __restrict__ rows,
                                                           const float* __restrict__ rows1, float* __restrict__ emb,
                                                           float* __restrict__ lr1) {
    const int f = blockIdx.y;
    const int grp = threadIdx.x / C, cc = threadIdx.x - grp * C;
    const int gpb = 256 / C;
    const int i = blockIdx.x * gpb + grp;
    if (grp >= gpb || i >= counts[f * 4 + 0]) return;
    const long long src = sorted_src[(long long)f * cap + i];
    const int u = entry_seg[(long long)f * cap + i];
    long long pos = -1;
    if (u >= 0) pos = route[(long long)f * cap + u];          // -1: the key did not fit its bucket
    if (emb) {
        if constexpr (VEC == 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos >= 0) v = *reinterpret_cast<const float4*>(rows + pos * d + cc * 4);
            *reinterpret_cast<float4*>(emb + src * d + cc * 4) = v;
        } else {
            emb[src * d + cc] = pos >= 0 ? rows[pos * d + cc] : 0.f;
        }
    }
    if (cc == 0 && lr1) lr1[src] = pos >= 0 ? rows1[pos] : 0.f;
}

// Requester side: the gradient of each distinct key into its bucket position.
template <int VEC>
__global__ __launch_bounds__(256) void shard_pack_kernel(const int* __restrict__ counts, const int* __restrict__ route,
                                                         int F, int cap, int d, int C,
                                                         const float* __restrict__ rowgrad,
                                                         const float* __restrict__ rowgrad1, float* __restrict__ grads,
                                                         float* __restrict__ grads1, int ld, int ld1) {
    const int f = blockIdx.y;
    const int grp = threadIdx.x / C, cc = threadIdx.x - grp * C;
    const int gpb = 256 / C;
    const int u = blockIdx.x * gpb + grp;
    if (grp >= gpb || u >= counts[f * 4 + 2]) return;
    const long long pos = route[(long long)f * cap + u];
    if (pos < 0) return;
    const long long g = (long long)f * cap + u;
    if (grads) {
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(grads + pos * ld + cc * 4) = *reinterpret_cast<const float4*>(rowgrad + g * d + cc * 4);
        } else {
            grads[pos * ld + cc] = rowgrad[g * d + cc];
        }
    }
    if (cc == 0 && grads1) grads1[pos * ld1] = rowgrad1[g];
}

// Owner side: sub-list descriptors for the merge sort from the count slots of the received buckets -- and, on the
// workgroups behind those, the requester's slot map (slot_map_entry; grid.x = count blocks + F * ceil(cap / 256)):
// two small pieces of bookkeeping that both have to be there before rows are served and looked up.
__global__ __launch_bounds__(256) void shard_counts_kernel(const long long* __restrict__ recv_idx, int N, int F,
                                                           const int* __restrict__ capf, const int* __restrict__ boff, int S,
                                                           int* __restrict__ sub_cnt, int count_blocks,
                                                           const int* __restrict__ sorted_src,
                                                           const int* __restrict__ entry_seg,
                                                           const int* __restrict__ counts, const int* __restrict__ route,
                                                           int cap, int* __restrict__ slot_map, int B, int Fdiv) {
    if ((int)blockIdx.x >= count_blocks) {
        const int per = (cap + 255) / 256, b = (int)blockIdx.x - count_blocks;
        slot_map_entry(sorted_src, entry_seg, counts, route, cap, N * S, slot_map, B, Fdiv, b / per,
                       (b % per) * 256 + (int)threadIdx.x);
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * F) {
        const int src = i / F, f = i - src * F;
        const long long c = recv_idx[(long long)src * (S + F) + boff[f] + f];
        const int cap_b = capf[f];
        sub_cnt[i] = (int)(c < 0 ? 0 : (c > cap_b ? cap_b : c));
    }
}

// C lanes per row, 256 / C rows per workgroup: rows of up to 1024 floats (field-aware FFM rows) take a whole workgroup.
int lanes(int d, int* VEC, int* C) {
    if (d % 4 == 0 && d / 4 <= 256) { *VEC = 4; *C = d / 4; return 0; }
    if (d <= 64) { *VEC = 1; *C = d; return 0; }
    return -1;
}

}  // namespace

extern "C" {

// Requester: bucket the distinct keys of a sorted batch by owning rank (key % N); see the file header.
// plan: the rs_sparse_bwd_sort plan of the local batch (capacity cap per field).  capf / boff: device int32 [F].
// send_idx: int64 [N, S + F]; route: int32 [F, cap].
int rs_shard_route(const int* plan, int F, int cap, const int64_t* row_offset, int N, const int* capf, const int* boff,
                   int S, long long* send_idx, int* route, int* overflow_flag, hipStream_t stream) {
    RS_CHECK_ARG(plan && row_offset && send_idx && route && capf && boff, "rs_shard_route: null argument");
    RS_CHECK_ARG(F > 0 && cap > 0 && N > 0 && S > 0, "rs_shard_route: bad sizes");
    RS_CHECK_ARG((long long)N * S < (1ll << 31), "rs_shard_route: %d x %d bucket slots do not fit an int", N, S);
    RS_CHECK_ARG(cap <= RT_THREADS * 16, "rs_shard_route: %d entries per field exceed %d", cap, RT_THREADS * 16);
    const PlanView pv = carve_plan(const_cast<int*>(plan), F, cap);
    int it = 1;
    while (it * RT_THREADS < cap) it <<= 1;
#define RS_ROUTE(IT)                                                                                         \
    hipLaunchKernelGGL(route_kernel<IT>, dim3(F), dim3(RT_THREADS), 0, stream, pv.seg_row, pv.counts, row_offset, F, \
                       cap, N, capf, boff, S, send_idx, route, overflow_flag)
    if (it <= 4) RS_ROUTE(4);
    else if (it == 8) RS_ROUTE(8);
    else RS_ROUTE(16);
#undef RS_ROUTE
    RS_CHECK_LAUNCH("rs_shard_route");
    return RS_OK;
}

// Owner: look up the rows named by the received buckets.  rows: [N, S, ld] (ld >= d floats per position); rows1: the
// 1-dim values, ld1 floats apart (a spare column of the same buffer, or [N, S] with ld1 = 1); cap_max = max capf.
int rs_shard_gather(const long long* recv_idx, const int64_t* local_row_offset, const int64_t* local_vocab, int N,
                    int F, const int* capf, const int* boff, int S, int cap_max, int d, const float* table,
                    const float* table1, float* rows, float* rows1, int ld, int ld1, int* err_flag, hipStream_t stream) {
    RS_CHECK_ARG(recv_idx && local_row_offset && local_vocab && capf && boff, "rs_shard_gather: null argument");
    RS_CHECK_ARG(N > 0 && F > 0 && S > 0 && cap_max > 0, "rs_shard_gather: bad sizes");
    RS_CHECK_ARG((table == nullptr) == (rows == nullptr) && (table1 == nullptr) == (rows1 == nullptr),
                 "rs_shard_gather: table/rows mismatch");
    int VEC, C;
    RS_CHECK_ARG(lanes(d, &VEC, &C) == 0, "rs_shard_gather: unsupported d=%d", d);
    RS_CHECK_ARG((!rows || ld >= d) && (!rows1 || ld1 >= 1), "rs_shard_gather: bad row strides");
    if (VEC == 4 && rows && (ld % 4 != 0 || (reinterpret_cast<uintptr_t>(rows) & 15))) { VEC = 1; C = d; }
    RS_CHECK_ARG(C <= 256, "rs_shard_gather: rows of %d floats need 16-byte aligned positions", d);
    dim3 grid(rs_cdiv(cap_max, 256 / C), N * F), block(256);
    if (VEC == 4) hipLaunchKernelGGL(shard_gather_kernel<4>, grid, block, 0, stream, recv_idx, local_row_offset, local_vocab,
                                     F, capf, boff, S, d, C, table, table1, rows, rows1, ld, ld1, err_flag);
    else hipLaunchKernelGGL(shard_gather_kernel<1>, grid, block, 0, stream, recv_idx, local_row_offset, local_vocab, F,
                            capf, boff, S, d, C, table, table1, rows, rows1, ld, ld1, err_flag);
    RS_CHECK_LAUNCH("rs_shard_gather");
    return RS_OK;
}

// Requester: expand the returned distinct rows to the batch.  emb: [B*F, d] (gradient-row order b*F+f).
int rs_shard_expand(const int* plan, const int* route, int F, int cap, int d, const float* rows,
                    const float* rows1, float* emb, float* lr1, hipStream_t stream) {
    RS_CHECK_ARG(plan && route, "rs_shard_expand: null argument");
    RS_CHECK_ARG((rows == nullptr) == (emb == nullptr) && (rows1 == nullptr) == (lr1 == nullptr),
                 "rs_shard_expand: rows/emb mismatch");
    int VEC, C;
    RS_CHECK_ARG(lanes(d, &VEC, &C) == 0, "rs_shard_expand: unsupported d=%d", d);
    const PlanView pv = carve_plan(const_cast<int*>(plan), F, cap);
    dim3 grid(rs_cdiv(cap, 256 / C), F), block(256);
    if (VEC == 4) hipLaunchKernelGGL(shard_expand_kernel<4>, grid, block, 0, stream, pv.sorted_src, pv.entry_seg, pv.counts,
                                     route, F, cap, d, C, rows, rows1, emb, lr1);
    else hipLaunchKernelGGL(shard_expand_kernel<1>, grid, block, 0, stream, pv.sorted_src, pv.entry_seg, pv.counts, route,
                            F, cap, d, C, rows, rows1, emb, lr1);
    RS_CHECK_LAUNCH("rs_shard_expand");
    return RS_OK;
}

// Requester, fused form of the expansion for rows the fused gather takes (d % 4 == 0 and d <= 256, or d <= 64): K1 --
// rs_gather_lr_fm_fwd, same outputs, same arithmetic -- reading entry (f, b)'s row at position slot_map[f*B + b] of the
// received row buffer (rows: ld floats per position; rows1: the 1-dim values, ld1 floats apart; either may be null).
int rs_shard_lookup(const int* slot_map, const float* rows, int ld, const float* rows1, int ld1, const float* bias,
                    float* emb_out, float* sum_out, float* bi_out, float* lr_out, float* fm_out, int B, int F, int d,
                    hipStream_t stream) {
    if (B == 0) return RS_OK;
    RS_CHECK_ARG(slot_map, "rs_shard_lookup: null slot map");
    RS_CHECK_ARG(B > 0 && F > 0 && d > 0, "rs_shard_lookup: bad sizes B=%d F=%d d=%d", B, F, d);
    RS_CHECK_ARG(rows || rows1, "rs_shard_lookup: both row buffers are null");
    RS_CHECK_ARG(rows || !(emb_out || sum_out || bi_out || fm_out), "rs_shard_lookup: embedding outputs requested without rows");
    RS_CHECK_ARG(rows1 || !lr_out, "rs_shard_lookup: lr_out requested without rows1");
    RS_CHECK_ARG((!rows || ld >= d) && (!rows1 || ld1 >= 1), "rs_shard_lookup: bad row strides");
    GatherArgs a{nullptr, nullptr, nullptr, rows, rows1, bias, emb_out, sum_out, bi_out, lr_out, fm_out, nullptr, B, F, d, 1, 1,
                 slot_map, ld, ld1};
    int VEC = 1;
    if (!rows) { VEC = 1; a.C = 1; a.G = 8; a.d = 1; }
    else {
        RS_CHECK_ARG(pick_layout(d, &VEC, &a.G, &a.C) == 0, "rs_shard_lookup: unsupported embedding_dim %d", d);
        if (VEC == 4 && (ld % 4 != 0 || (reinterpret_cast<uintptr_t>(rows) & 15))) {
            RS_CHECK_ARG(d <= 64, "rs_shard_lookup: rows of %d floats need 16-byte aligned positions", d);
            VEC = 1; a.C = d; a.G = 8;
            while (a.G > 1 && a.G * a.C > 32) a.G >>= 1;
            RS_CHECK_ARG(a.G * a.C <= 64, "rs_shard_lookup: unsupported embedding_dim %d", d);
        }
    }
    const int spw = 64 / (a.G * a.C);
    const int waves = rs_cdiv(B, spw);
    dim3 grid(rs_cdiv(waves, 4)), block(256);
    if (VEC == 4) hipLaunchKernelGGL(shard_lookup_kernel<4>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(shard_lookup_kernel<1>, grid, block, 0, stream, a);
    RS_CHECK_LAUNCH("rs_shard_lookup");
    return RS_OK;
}

// Requester: distinct-key gradients (rowgrad [F,cap,d], rowgrad1 [F,cap]) into bucket order (grads: ld floats per
// position; grads1: ld1 floats apart).
int rs_shard_pack(const int* plan, const int* route, int F, int cap, int d, const float* rowgrad,
                  const float* rowgrad1, float* grads, float* grads1, int ld, int ld1, hipStream_t stream) {
    RS_CHECK_ARG(plan && route, "rs_shard_pack: null argument");
    RS_CHECK_ARG((rowgrad == nullptr) == (grads == nullptr) && (rowgrad1 == nullptr) == (grads1 == nullptr),
                 "rs_shard_pack: rowgrad/grads mismatch");
    int VEC, C;
    RS_CHECK_ARG(lanes(d, &VEC, &C) == 0, "rs_shard_pack: unsupported d=%d", d);
    RS_CHECK_ARG((!grads || ld >= d) && (!grads1 || ld1 >= 1), "rs_shard_pack: bad row strides");
    if (VEC == 4 && grads && (ld % 4 != 0 || (reinterpret_cast<uintptr_t>(grads) & 15))) { VEC = 1; C = d; }
    RS_CHECK_ARG(C <= 256, "rs_shard_pack: rows of %d floats need 16-byte aligned positions", d);
    const PlanView pv = carve_plan(const_cast<int*>(plan), F, cap);
    dim3 grid(rs_cdiv(cap, 256 / C), F), block(256);
    if (VEC == 4) hipLaunchKernelGGL(shard_pack_kernel<4>, grid, block, 0, stream, pv.counts, route, F, cap, d, C,
                                     rowgrad, rowgrad1, grads, grads1, ld, ld1);
    else hipLaunchKernelGGL(shard_pack_kernel<1>, grid, block, 0, stream, pv.counts, route, F, cap, d, C, rowgrad,
                            rowgrad1, grads, grads1, ld, ld1);
    RS_CHECK_LAUNCH("rs_shard_pack");
    return RS_OK;
}

// Owner: sub_cnt[s*F+f] = count slot of received bucket (s, f), clamped to [0, capf[f]].  In the same launch, when
// slot_map (int32 [F, B]) is given: the requester's map from every entry (f, b) of its sorted batch (plan, route of
// rs_shard_route, capacity cap) to the position of its key's row among the received rows -- N*S, a row of zeros the
// caller keeps there, for padding entries and keys that did not fit their bucket; Fdiv: sample = gradient row / Fdiv.
int rs_shard_counts(const long long* recv_idx, int N, int F, const int* capf, const int* boff, int S, int* sub_cnt,
                    const int* plan, const int* route, int cap, int* slot_map, int B, int Fdiv, hipStream_t stream) {
    RS_CHECK_ARG(recv_idx && sub_cnt && capf && boff && N > 0 && F > 0 && S > 0, "rs_shard_counts: bad arguments");
    RS_CHECK_ARG(!slot_map || (plan && route && cap > 0 && B > 0 && B <= cap && Fdiv > 0), "rs_shard_counts: bad slot map arguments");
    const int cb = rs_cdiv(N * F, 256);
    PlanView pv{};
    if (slot_map) pv = carve_plan(const_cast<int*>(plan), F, cap);
    const int sb = slot_map ? F * rs_cdiv(cap, 256) : 0;
    hipLaunchKernelGGL(shard_counts_kernel, dim3(cb + sb), dim3(256), 0, stream, recv_idx, N, F, capf, boff, S, sub_cnt, cb,
                       pv.sorted_src, pv.entry_seg, pv.counts, route, cap, slot_map, B, Fdiv);
    RS_CHECK_LAUNCH("rs_shard_counts");
    return RS_OK;
}

}  // extern "C"
