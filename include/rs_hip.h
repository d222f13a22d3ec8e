/*
 * librs_hip.so -- C-ABI of the MI355X (gfx950) CTR training hot path.
 *
 * The reference (HaochengY/RecommendSystem) is pure Python/PyTorch and has no FFI: the interface it
 * exposes for this path is the nn.Module surface of model/utils/Layers.py as driven by
 * model/utils/BaseModel.py.  Each entry point below therefore cites the reference lines whose
 * arithmetic it replaces; the Python modules in recommendsystem_amd/ keep the reference's class
 * names, constructor arguments and forward signatures and call these functions through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; fp32 data, int64 indices
 *     (model/utils/Layers.py:147 `.long()`), int32 scratch;
 *   - functions return 0 on success, a negative code on error (-1 argument/shape, -2 launch,
 *     -3 workspace); rs_last_error_string() gives the message of the calling thread's last error;
 *   - nothing is allocated, freed or synchronised inside: outputs and scratch are caller-owned, work is
 *     enqueued on `stream` (safe under HIP graph capture);
 *   - index batches are columnar: idx[f*B + b] is the index of field f in sample b, i.e. the
 *     reference's dict of per-column (B,) tensors laid end to end;
 *   - all per-field tables live in one stacked table [R, d]; row_offset[f] is field f's first row,
 *     vocab[f] its row count; row 0 of every field is the padding row (padding_idx=0);
 *   - activation codes: 0 none, 1 ReLU, 2 Tanh, 3 LeakyReLU(0.01), 4 PReLU;
 *     optimizer codes: 0 SGD, 1 Adam, 2 AdamW, 3 Adagrad.
 */
#ifndef RS_HIP_H
#define RS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* rs_stream_t; /* hipStream_t */

/* ---- library ------------------------------------------------------------------------------- */
int rs_version(void);
const char* rs_last_error_string(void);
int rs_device_cu_count(void);

/* ---- K1: fused multi-field gather ----------------------------------------------------------- */
/* out[b,f,:] = table[row_offset[f] + idx[f,b], :]
 * replaces Embedding_layer.forward: one nn.Embedding call per column + torch.stack
 * (model/utils/Layers.py:137-167).  Out-of-range indices read the padding row and set *err_flag. */
int rs_gather_fwd(const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, const float* table,
                  float* out, int B, int F, int d, int* err_flag, rs_stream_t stream);

/* One pass over the rows producing any of: emb_out [B,F,d]; sum_out [B,d] = sum_f e_f;
 * bi_out [B,d] = 0.5*((sum_f e)^2 - sum_f e^2) (NFM Bi-Interaction, textbook form);
 * lr_out [B] = sum_f table1[row] + bias[0]  (LinearRegression_layer.forward, Layers.py:107-109);
 * fm_out [B] = 0.5*sum_k((sum_f e)^2 - sum_f e^2)  (model/FM/FM.py:27-30, model/DeepFM/DeepFM.py:47-49).
 * Null outputs are skipped; table or table1 may be null when unused. */
int rs_gather_lr_fm_fwd(const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, const float* table,
                        const float* table1, const float* bias, float* emb_out, float* sum_out, float* bi_out,
                        float* lr_out, float* fm_out, int B, int F, int d, int* err_flag, rs_stream_t stream);

/* rs_gather_lr_fm_fwd on tables kept as w / *scale (scale: one float in device memory; null = 1): every looked-up
 * value, and through them the sums, the LR logit without its bias and the FM term, is that of the table times *scale.
 * For trainers that apply the per-step shrink of the L2 term (BaseModel.py:214-231 under SGD: every row of every
 * nn.Embedding is multiplied by 1 - lr c lambda per step) to one scalar instead of every row. */
int rs_gather_lr_fm_fwd_scaled(const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, const float* table,
                               const float* table1, const float* bias, float* emb_out, float* sum_out, float* bi_out,
                               float* lr_out, float* fm_out, int B, int F, int d, const float* scale, int* err_flag,
                               rs_stream_t stream);

/* ---- K2e: MLP GEMMs on the matrix cores (fp32 in, fp32 accumulate) --------------------------- */
/* C[M,N] = epi(op(A) op(B)),  epi(v) = act(v + bias[n]) * act'(mulsrc[m,n]) (each part optional).
 * a_mc = 0: A is [M,K] row-major; 1: A is stored [K,M].  b_nc = 0: B is [N,K] (nn.Linear weight);
 * 1: B is stored [K,N].  Replaces nn.Linear forward and its autograd backward inside MLPBlock /
 * ResidualMLPBlock (model/utils/Layers.py:39-47, 84-94):
 *   forward  H  = act(X W^T + b)          (a_mc=0, b_nc=0)
 *   dX       dZ'= (dZ W) * act'(H_prev)   (a_mc=0, b_nc=1)
 *   dW       dW = dZ^T X                  (a_mc=1, b_nc=1, split over the batch when splitk > 1)
 * tile < 0 / splitk <= 0: chosen by rs_gemm_plan.  workspace is needed when the split count > 1.
 * colsum_part (optional; needs an explicit tile and an unsplit reduction): [rs_gemm_colsum_rows(M, tile), N] partial
 *   column sums of the finished C, one row per wave row of the grid.  Their sum over rows is sum_m C[m,n] -- the bias
 *   gradient db = sum_b dZ[b,:] when C is a dZ (nn.Linear's bias backward); rs_colsum_finish_multi adds up the
 *   partials of several GEMMs (and of rs_head_bwd) in one launch. */
size_t rs_gemm_workspace_bytes(int M, int N, int splitk);
int rs_gemm_colsum_rows(int M, int tile);
int rs_gemm_plan(int M, int N, int K, int cus, int* splitk_out /* host pointer */);
int rs_gemm_f32(int a_mc, int b_nc, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                int ldc, const float* bias, int act, const float* mulsrc, int ldm, int mul_act, int tile, int splitk,
                float* workspace, size_t workspace_bytes, float* colsum_part, rs_stream_t stream);
/* rs_gemm_f32 with the mask of rs_dropout_apply(drop_p, drop_seed, drop_salt, drop_call_host, drop_call_dev) multiplied
 * into the result (element index m*N + n) after bias and activation and before the mulsrc factor: MLPBlock's
 * dropout(act(fc(x))) (Layers.py:39-47) in the forward product's epilogue, and in the dX product that feeds such a
 * block's backward.  drop_p == 0: plain rs_gemm_f32. */
int rs_gemm_f32_drop(int a_mc, int b_nc, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                     int ldc, const float* bias, int act, const float* mulsrc, int ldm, int mul_act, int tile, int splitk,
                     float* workspace, size_t workspace_bytes, float* colsum_part, float drop_p,
                     unsigned long long drop_seed, unsigned drop_salt, unsigned long long drop_call_host,
                     const long long* drop_call_dev, rs_stream_t stream);
/* C = act(C + op(A) op(B) + bias): one term of a product whose reduction dimension is walked in several calls, so that
 * the whole left operand never has to exist at once -- the first MLP layer of PNN over the outer products of all field
 * pairs (model/PNN/PNN.py:19-27,58,80-103: 741 * 256 inputs per sample at F = 39, d = 16), evaluated a block of pairs at
 * a time.  bias and act belong to the last call. */
int rs_gemm_f32_acc(int a_mc, int b_nc, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                    int ldc, const float* bias, int act, int tile, int splitk, float* workspace, size_t workspace_bytes,
                    rs_stream_t stream);
/* A split product without its reducing launch: the splitk slabs [splitk, M, N] stay in `workspace` until
 * rs_splitk_reduce_multi adds them, in slab order, into contiguous [M, N] outputs for up to 8 such products in one
 * launch (host arrays of device pointers / element counts / split counts).  Used for the weight gradients of a
 * backward pass (Layers.py:42 nn.Linear under autograd), which nothing reads before the gradient norm. */
int rs_gemm_f32_slabs(int a_mc, int b_nc, int M, int N, int K, const float* A, int lda, const float* B, int ldb, int tile,
                      int splitk, float* workspace, size_t workspace_bytes, rs_stream_t stream);
int rs_splitk_reduce_multi(const float* const* ws, float* const* C, const long long* total, const int* splitk, int count,
                           rs_stream_t stream);
/* Up to four independent products in ONE launch (64x64 tiles, `stages` = 2 or 3 LDS ring stages): a layer's dX and dW
 * both need only its dZ (nn.Linear's autograd backward, Layers.py:42), and nothing reads either before the next
 * launch, so their workgroups fill the chip together.  A descriptor with splitk == 1 is an rs_gemm_f32 product
 * (bias / act / mulsrc / colsum_part epilogue into C; colsum_part has ceil(M/64) rows); with splitk > 1 it is an
 * rs_gemm_f32_slabs product (slabs left in its workspace, C and the epilogue fields unused).  Operands must allow
 * 16-byte accesses.  The products must not read one another's results. */
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
int rs_gemm_f32_group(const rs_gemm_desc* descs /* host array */, int count, int stages, rs_stream_t stream);
/* rs_gemm_f32_group (3 ring stages) whose launch also carries rs_sparse_bwd_sort of a columnar batch idx [F, B]
 * (arguments as for rs_sort_gather; B <= rs_group_sort_max_entries()): the index bookkeeping that the backward of the
 * NEXT batch needs (Layers.py:137-160 under autograd) depends on nothing but that batch, keeps one workgroup per field
 * busy and fits the LDS of a GEMM workgroup, so it rides beside a block's backward products instead of standing in
 * front of a step.  count may be 0.  The plan written is rs_sparse_bwd_sort's, bit for bit.  phase 0: the whole sort;
 * phase 1 / 2: its first split_passes radix passes in one launch (items parked in the plan's scratch), the rest and
 * the plan in a later one -- beside GEMM workgroups a sort workgroup runs several times as long as alone. */
int rs_group_sort_max_entries(void);
int rs_gemm_f32_group_sort(const rs_gemm_desc* descs /* host array */, int count, const int64_t* idx, const int* sub_off,
                           const int* sub_cnt, const int* pay_base, int cap, int L, const int64_t* vocab,
                           const int64_t* row_offset, int* plan, int B, int F, int phase, int split_passes,
                           rs_stream_t stream);
/* out[p][n] = sum_{r < rows[p]} part[p][r*ld[p] + n] for n < n[p], p < count <= 8, in one launch.  The five arrays
 * are HOST arrays (of device pointers / sizes). */
int rs_colsum_finish_multi(const float* const* part, const int* rows, const int* n, const int* ld, float* const* out,
                           int count, rs_stream_t stream);

/* out[n] = sum_m scale[m] * X[m,n] (scale optional): bias gradient of nn.Linear, dgamma/dbeta sums of
 * LayerNorm, and (with scale) the weight gradient of a 1-wide layer. */
size_t rs_colsum_workspace_bytes(int M, int N);
int rs_colsum_f32(const float* X, int M, int N, int ld, const float* scale, float* out, float* workspace,
                  size_t workspace_bytes, rs_stream_t stream);
/* Stage 1 of rs_colsum_f32 alone: part [rs_colsum_partial_rows(M), N] partial column sums, to be finished together
 * with those of other producers by one rs_colsum_finish_multi launch (the bias gradients of a backward pass that
 * nothing reads before the gradient norm). */
int rs_colsum_partial_rows(int M);
int rs_colsum_partials(const float* X, int M, int N, int ld, const float* scale, float* part, rs_stream_t stream);
/* rs_colsum_f32 without a workspace argument, for callers that cannot allocate: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  The stage-1 partial sums
 * go to a device array the library owns (4 MiB on every GPU the library runs on, held until the process ends):
 * ceil(M/64) * N must not exceed rs_colsum_noalloc_max_floats() = 2^20, larger problems are rejected with -1.  The
 * call allocates and frees nothing.  The first use on a device looks up that device's address of the array -- a
 * runtime query, not a stream operation: make it outside any stream capture, either with one ordinary call or with
 * rs_colsum_noalloc_prepare() (device current), after which the call may be captured.  Every call on a GPU shares
 * that GPU's array: calls that overlap on one GPU (different streams) race. */
size_t rs_colsum_noalloc_max_floats(void);
int rs_colsum_noalloc_prepare(void);
int rs_colsum_f32_noalloc(const float* X, int M, int N, int ld, const float* scale, float* out, rs_stream_t stream);

/* 1-wide linear layer (the MLP head, model/DeepFM/DeepFM.py:35): h[m] = act(X[m,:].w + bias[0]) and
 * dX[m,k] = dz[m] * w[k] * act'(X[m,k]). */
int rs_rowdot_fwd(const float* X, int M, int K, const float* w, const float* bias, int act, float* h,
                  rs_stream_t stream);
int rs_rowdot_bwd_dx(const float* dz, const float* w, const float* X, int mul_act, int M, int K, float* dX,
                     rs_stream_t stream);

/* activation / normalisation layers of MLPBlock (Layers.py:14-29, 43-45) */
int rs_act_fwd(const float* z, float* h, long long n, int act, const float* slope, rs_stream_t stream);
int rs_act_bwd_num_partials(long long n);
int rs_act_bwd(const float* dh, const float* z, const float* h, float* dz, long long n, int act, const float* slope,
               float* dslope_partials, rs_stream_t stream);
int rs_bn_fwd(const float* z, int M, int N, const float* gamma, const float* beta, float eps, int act, int training,
              float momentum, float* mean, float* var, float* running_mean, float* running_var, float* xhat, float* h,
              rs_stream_t stream);
int rs_bn_bwd(const float* dh, const float* h, const float* xhat, int M, int N, int act, const float* gamma,
              const float* var, float eps, float* dgamma, float* dbeta, float* dz, rs_stream_t stream);
/* BatchNorm1d over a batch that is spread over several GPUs (no counterpart in the reference, which is single-device;
 * the arithmetic is its BatchNorm1d on the union batch, Layers.py:27,44): the pieces of the two calls above, between
 * which the caller combines the ranks' results.  rs_bn_stats: this rank's mean and biased variance; the normalisation
 * is rs_bn_fwd(training = 0) with the combined statistics; rs_bn_bwd_reduce: this rank's dbeta / dgamma sums;
 * rs_bn_bwd_apply: dz from the sums over all m_total rows of the union batch. */
int rs_bn_stats(const float* z, int M, int N, float* mean, float* var, rs_stream_t stream);
int rs_bn_bwd_reduce(const float* dh, const float* h, const float* xhat, int M, int N, int act, float* dgamma,
                     float* dbeta, rs_stream_t stream);
int rs_bn_bwd_apply(const float* dh, const float* h, const float* xhat, int M, int N, int act, const float* gamma,
                    const float* var, float eps, const float* dgamma_all, const float* dbeta_all, int m_total, float* dz,
                    rs_stream_t stream);
int rs_ln_fwd(const float* z, int M, int N, const float* gamma, const float* beta, float eps, int act, float* xhat,
              float* rstd, float* h, rs_stream_t stream);
int rs_ln_bwd(const float* dh, const float* h, const float* xhat, const float* rstd, const float* gamma, int M, int N,
              int act, float* dz, float* dy, float* dyxhat, rs_stream_t stream);
int rs_add_f32(const float* a, const float* b, float* y, long long n, rs_stream_t stream);   /* residual add, Layers.py:94 */
/* nn.Dropout in training mode (Layers.py:33-37,46): y[i] = keep(i) ? x[i]/(1-p) : 0.  keep(i) comes from Philox4x32-10
 * keyed by `seed` with counter (i/4, call, salt): element i's bit depends only on (seed, salt, call, i), so calling it
 * again on the gradient with the same four numbers is the backward pass and no mask is stored.  call = call_host +
 * *call_dev (call_dev, a device counter such as the optimizer's step count, may be null): a replayed HIP graph then
 * draws a fresh mask per replay.  The reference's masks come from torch's generator; only the distribution matches. */
int rs_dropout_apply(const float* x, float* y, long long n, float p, unsigned long long seed, unsigned salt,
                     unsigned long long call_host, const long long* call_dev, rs_stream_t stream);
int rs_axpy_f32(float a, const float* x, float* y, long long n, rs_stream_t stream);         /* L2 term, BaseModel.py:214-231 */
int rs_loss_add_reg(float* loss, const float* partials, const int* seg_end, const float* coef, int nseg,
                    rs_stream_t stream);

/* ---- loss head: sigmoid + F.binary_cross_entropy + dL/dlogit (BaseModel.py:145, LR.py:32) --- */
int rs_bce_head(const float* z0, const float* z1, const float* z2, const float* labels_f32, const int64_t* labels_i64,
                int B, float* logit_out, float* prob_out, float* loss_out, float* dz_out, float grad_scale,
                float* sum_dz_out, const float* hlast, int act_last, float* dzh_out, float* sum_dzh_out,
                rs_stream_t stream);

/* The same head with the 1-wide last MLP block (an MLPBlock(h_last, 1): model/DeepFM/DeepFM.py:35) fused in:
 * h = act(X.w + b), logit = z0 + z1 + h (z0, z1 nullable), loss, dz, dzh = dz*act'(h) and their sums in one launch;
 * and its backward dX = dzh w^T * act'_prev(X) plus part [rs_head_bwd_rows(B), 2K]: partial sums of
 * [dW = X^T dzh | dbp = column sums of dX], to be added over rows by rs_colsum_finish_multi.
 * ws: rs_head_workspace_floats(B, K) floats; sem: one int, zero before the first call, left zero.
 * sem == NULL: the three batch sums are left to the caller: ws[4g + {0,1,2}], g < ceil(B/16), holds loss/B, sum dz
 * and sum dzh of 16 rows each, to be added over g (rs_colsum_finish_multi, ld = 4); loss_out, sum_dz_out and
 * sum_dzh_out are then not written. */
size_t rs_head_workspace_floats(int B, int K);
int rs_head_fwd_bce(const float* X, int B, int K, const float* w, const float* bias, int act, const float* z0,
                    const float* z1, const float* labels_f32, const int64_t* labels_i64, float grad_scale, float* h_out,
                    float* loss_out, float* dz_out, float* sum_dz_out, float* dzh_out, float* sum_dzh_out, float* ws,
                    int* sem, rs_stream_t stream);
/* rs_head_fwd_bce in its sem == NULL form and rs_head_bwd in one launch (X is read once): h, dz, dzh, the batch-sum
 * partials in ws, dX [B, K] and part [rs_head_bwd_rows(B), 2K].  K % 4 == 0, K <= 1024.  (DeepFM.py:35 forward and
 * backward of the 1-wide block + BaseModel.py:145.) */
int rs_head_fwd_bwd(const float* X, int B, int K, const float* w, const float* bias, int act, const float* z0,
                    const float* z1, const float* labels_f32, const int64_t* labels_i64, float grad_scale, float* h_out,
                    float* dz_out, float* dzh_out, float* ws, int mul_act, float* dX, float* part, rs_stream_t stream);
int rs_head_bwd_rows(int B);
int rs_head_bwd(const float* dzh, const float* w, const float* X, int mul_act, int B, int K, float* dX, float* part,
                rs_stream_t stream);
/* The 1-wide layer over cat([X1, X2], 1) without the concatenation -- DCN's Linear(D + h_last, 1) over the cross
 * network's output and the deep tower's (model/DCN/DCN.py:31,52): h[m] = act(X1[m,:].w[:K1] + X2[m,:].w[K1:] + b);
 * the backward is rs_head_bwd with dX and the columns of part [rs_head_bwd_rows(B), 2*(K1+K2)] split the same way,
 * mul_act2 = the activation whose output X2 is (0: none).  K1 + K2 <= 1024. */
int rs_rowdot2_fwd(const float* X1, int K1, const float* X2, int K2, int M, const float* w, const float* bias, int act,
                   float* h, rs_stream_t stream);
int rs_head2_bwd(const float* dzh, const float* w, const float* X1, int K1, const float* X2, int K2, int mul_act2, int B,
                 float* dX1, float* dX2, float* part, rs_stream_t stream);

/* ---- K2: interactions on gathered embeddings emb[B,F,d] -------------------------------------- */
int rs_fm_fwd(const float* emb, int B, int F, int d, float* sum_out, float* bi_out, float* fm_out, rs_stream_t stream);
int rs_fm_bwd(const float* emb, const float* S, const float* dfm, const float* dbi, int B, int F, int d, float* demb,
              rs_stream_t stream);

/* Pairs (i<j) are numbered row-major, p = i*(2F-i-1)/2 + (j-i-1), as the reference's nested loops emit them.
 * The *_bwd entry points that produce parameter gradients write per-workgroup partial rows
 * (rows = rs_interact_partial_rows(B)); the caller sums them over rows with rs_colsum_f32. */
int rs_interact_partial_rows(int B);

/* PNN inner products <e_i,e_j> -> out[B,P]                       (model/PNN/PNN.py:65-77) */
int rs_pnn_inner_fwd(const float* emb, int B, int F, int d, float* out, rs_stream_t stream);
int rs_pnn_inner_bwd(const float* emb, const float* dout, int B, int F, int d, float* demb, rs_stream_t stream);
/* The input row of PNN's first MLP layer with pnn_type inner, cat([emb.flatten(1), inner products], 1) (PNN.py:58),
 * written by the product kernel itself: out [B, ld], row = [F*d embedding values | P products | zeros], ld >= F*d + P
 * a multiple of four (the GEMM's operand alignment) with fewer than 64 columns of padding.  The backward takes the
 * gradient of such rows (row stride ld) and returns demb = dout[:, :F*d] + the products' share.
 * rs_pnn_inner_cat_ok: 1 when (F, d) is served (F <= 64, d in {4, 8, 16, 32}; emb and out 16-byte aligned). */
int rs_pnn_inner_cat_ok(int F, int d);
int rs_pnn_inner_cat_fwd(const float* emb, int B, int F, int d, float* out, long long ld, rs_stream_t stream);
int rs_pnn_inner_cat_bwd(const float* emb, const float* dout, long long ld, int B, int F, int d, float* demb,
                         rs_stream_t stream);
/* PNN outer products e_i e_j^T flattened -> out[B,P*d*d]         (model/PNN/PNN.py:80-103) */
int rs_pnn_outer_fwd(const float* emb, int B, int F, int d, float* out, rs_stream_t stream);
int rs_pnn_outer_bwd(const float* emb, const float* dout, int B, int F, int d, float* demb, rs_stream_t stream);
/* The same for the pairs (i, j) with i in [i_lo, i_hi) only (pairs are numbered first field major, PNN.py:83-85): out /
 * dout hold those pairs, (p_hi - p_lo)*d*d columns with row stride ld; with accumulate the backward adds to demb.  Lets
 * the first MLP layer over the products (PNN.py:19-27,58) run a block of pairs at a time (rs_gemm_f32_acc). */
int rs_pnn_outer_fwd_range(const float* emb, int B, int F, int d, int i_lo, int i_hi, float* out, long long ld,
                           rs_stream_t stream);
int rs_pnn_outer_bwd_range(const float* emb, const float* dout, long long ld, int B, int F, int d, int i_lo, int i_hi,
                           int accumulate, float* demb, rs_stream_t stream);
/* The first MLP layer of PNN over [x_other | outer products of all pairs] (model/PNN/PNN.py:19-27,58 fed by
 * PNN.py:80-103) without the (B, P*d*d) matrix of products:
 *   fwd  z[B,N] = [x_other | outer(emb)] W^T + bias        W is nn.Linear's weight [N, K0 + P*d*d], x_other [B,K0]
 *                                                           (row stride ldx; K0 = 0: none), emb [B,F,d]
 *   bwd  dW [N, K0 + P*d*d], db [N] (optional), demb [B,F,d], dx_other [B,K0] (optional) from dz [B,N]
 * The reduction dimension is walked a block of pairs at a time (rs_pnn_outer_fwd_range into a cache-sized buffer,
 * rs_gemm_f32_acc into the running sum); the backward recomputes each block.  workspace: 16-byte aligned device
 * memory of rs_pnn_outer_mlp_workspace_floats(B, F, d, K0, N, backward) floats (aligned copies of W and x_other, the
 * block buffers, split-K scratch; for the backward also dW in the aligned layout); nothing is kept between the two
 * calls.  block_bytes: size aimed at for one block of products (<= 0: 192 MiB, which stays in the 256 MB Infinity
 * Cache); rs_pnn_outer_mlp_blocks tells how many blocks that makes.  Activation, normalisation and dropout of the
 * MLPBlock are applied to z by the caller. */
size_t rs_pnn_outer_mlp_workspace_floats(int B, int F, int d, int K0, int N, int backward, long long block_bytes);
int rs_pnn_outer_mlp_blocks(int B, int F, int d, long long block_bytes);
int rs_pnn_outer_mlp_fwd(const float* emb, const float* x_other, long long ldx, int B, int F, int d, int K0, const float* W,
                         const float* bias, int N, float* z, float* workspace, size_t workspace_floats,
                         long long block_bytes, rs_stream_t stream);
int rs_pnn_outer_mlp_bwd(const float* emb, const float* x_other, long long ldx, const float* dz, int B, int F, int d, int K0,
                         const float* W, int N, float* dW, float* db, float* demb, float* dx_other, float* workspace,
                         size_t workspace_floats, long long block_bytes, rs_stream_t stream);

/* DCN cross network x_{l+1} = x0 (x_l . w_l) + b_l + x_l, w/bias [L,D]; S[B,L] = x_l . w_l saved for the backward
 * (model/DCN/DCN.py:48-52, with the bias broadcast over the batch).  partials: [rows, 2*L*D] = [dW | dB]. */
int rs_dcn_cross_fwd(const float* x0, const float* w, const float* bias, int B, int D, int L, float* out, float* S,
                     rs_stream_t stream);
int rs_dcn_cross_bwd(const float* x0, const float* S, const float* dout, const float* w, const float* bias, int B, int D,
                     int L, float* dx0, float* partials, int rows, rs_stream_t stream);
/* rs_dcn_cross_bwd with dx0 = (the cross network's gradient of x0) + dx_add [B, D] (the deep tower's gradient of the
 * same flattened embeddings, DCN.py:48-52); dx_add must not overlap dx0. */
int rs_dcn_cross_bwd_add(const float* x0, const float* S, const float* dout, const float* w, const float* bias, int B, int D,
                         int L, const float* dx_add, float* dx0, float* partials, int rows, rs_stream_t stream);

/* NFM bi-interaction as the reference writes it: 0.5*((E V)^2 - E^2 V^2), E[B,D], V[D,K], K <= 16
 * (model/NFM/NFM.py:42-44).  s_out[B,K] = E V is saved for the backward; partials: [rows, D*K] = dV. */
int rs_nfm_bi_fwd(const float* E, const float* V, int B, int D, int K, float* out, float* s_out, rs_stream_t stream);
int rs_nfm_bi_bwd(const float* E, const float* V, const float* s, const float* dout, int B, int D, int K, float* dE,
                  float* partials, int rows, rs_stream_t stream);

/* AFM: pairwise products, attention MLP (W[A,d], bA[A], h[A]), softmax over the pairs, weighted pooling and the
 * projection (pw[d], pb[1]) -> out[B]                             (model/AFM/AFM.py:19-24,36-42,47-56).
 * d in {4,8,16,32}; backward additionally needs A in {4,8,16} and A*d <= 256.
 * partials: [rows, A*d + 2A + d + 1] = [dW | dbA | dh | dpw | dpb]. */
int rs_afm_fwd(const float* emb, int B, int F, int d, int A, const float* W, const float* bA, const float* h,
               const float* pw, const float* pb, float* out, rs_stream_t stream);
int rs_afm_bwd(const float* emb, const float* dout, int B, int F, int d, int A, const float* W, const float* bA,
               const float* h, const float* pw, float* demb, float* partials, int rows, rs_stream_t stream);

/* FFM with one field-aware table per (feature, partner field), stored as T[sum V_f, F*d]; idx [F,B] columnar
 * (model/FFM/FFM.py:43-62; the reference aliases the F tables of a feature, this takes them independent).
 * rs_ffm_bwd writes the dense gradient dT (caller-zeroed) using the rs_sparse_bwd_sort plan of the same idx. */
int rs_ffm_fwd(const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, const float* T, int B, int F, int d,
               float* out, int* err_flag, rs_stream_t stream);
int rs_ffm_bwd(const int* plan, const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, const float* T,
               const float* dout, int B, int F, int cap, int d, float* dT, rs_stream_t stream);
/* The same gradient for the touched rows only: rowgrad [F, cap, F*d] holds segment s of field f of the plan at
 * [f, s, :] -- the layout rs_sparse_update / rs_sparse_scatter_dense read when called with row width F*d (rows of up
 * to 1024 floats) -- and norm_partials [rs_ffm_bwd_num_partials(F, cap)] the sums of squares for the global-norm clip
 * (BaseModel.py:148).  chunk_part [F, cap, F*d] is scratch (partial sums of rows that span several chunks of the
 * plan; the plan's window L must be <= 64).  Replaces the dense (V_f, F*d) gradient + dense optimizer sweep of
 * FFM.py's tables. */
int rs_ffm_bwd_num_partials(int F, int cap);
int rs_ffm_bwd_rows(const int* plan, const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, const float* T,
                    const float* dout, int B, int F, int cap, int d, float* rowgrad, float* chunk_part,
                    float* norm_partials, rs_stream_t stream);

/* FFM on rows already gathered by the row-sharded exchange: G [B, F, F*d] with G[b, i, :] the field-aware row of
 * sample b's value in field i (same arithmetic as rs_ffm_fwd, FFM.py:43-62); dG [B, F, F*d] is the gradient with
 * respect to the gathered rows, reduced per distinct key by rs_sparse_bwd_reduce_wide. */
int rs_ffm_rows_fwd(const float* G, int B, int F, int d, float* out, rs_stream_t stream);
int rs_ffm_rows_bwd(const float* G, const float* dout, int B, int F, int d, float* dG, rs_stream_t stream);

/* ---- K3: deterministic sparse embedding backward + optimizer (BaseModel.py:141-150) ---------- */
/* plan: int32 scratch of rs_sort_plan_ints(F, cap) elements.  Per field, `nsrc` sub-lists of keys are
 * concatenated (a columnar [F,B] batch is nsrc=1, sub_off[f]=f*B, sub_cnt[f]=B, pay_base[f]=f,
 * pay_stride=F so that entry (f,b) names gradient row b*F+f), stably radix-sorted in LDS, and cut into
 * segments (one per distinct non-padding row) and chunks (<= L entries, L = rs_sparse_window(d)). */
size_t rs_sort_plan_ints(int F, int cap);
int rs_sort_max_entries(void);
/* (Lists of more than 4096 entries given as one sub-list per field are sorted in two levels -- tiles, then a merge;
 * several sub-lists per field always take the one-workgroup sort.  The plan written is the same either way.) */
int rs_sparse_window(int d);
int rs_sparse_bwd_sort(const int64_t* keys, const int* sub_off, const int* sub_cnt, const int* pay_base, int pay_stride,
                       int nsrc, int F, int cap, int max_entries, int L, const int64_t* vocab,
                       const int64_t* row_offset, int* plan, rs_stream_t stream);
/* rs_sparse_bwd_sort for sub-lists that are each ascending already (the owner side of the row-sharded exchange
 * receives its keys that way): the same plan from a merge instead of a sort.  max_entries may reach
 * rs_merge_max_entries() (beyond rs_sort_max_entries()); nsrc <= 16.  No counterpart in the reference, which has
 * no multi-GPU path. */
int rs_merge_max_entries(void);
int rs_sparse_bwd_merge_lists(const int64_t* keys, const int* sub_off, const int* sub_cnt, const int* pay_base, int pay_stride,
                       int nsrc, int F, int cap, int max_entries, int L, const int64_t* vocab,
                       const int64_t* row_offset, int* plan, rs_stream_t stream);
/* rs_sparse_bwd_sort of a columnar batch idx [F, B] (sub_off[f] = f*B, sub_cnt[f] = B, pay_base[f] = f, stride F)
 * and rs_gather_lr_fm_fwd of the same batch in ONE launch: both need only the batch, and the sort occupies just F
 * workgroups, so the gather runs beside it (Layers.py:137-160 forward + the index bookkeeping of its backward).
 * resume_passes > 0: the sort was begun by rs_sort_catchup(..., split_passes = resume_passes) on the same plan and
 * batch; this launch runs its remaining radix passes and writes the plan. */
int rs_sort_gather(const int64_t* idx, const int* sub_off, const int* sub_cnt, const int* pay_base, int cap, int L,
                   const int64_t* vocab, const int64_t* row_offset, int* plan, const float* table, const float* table1,
                   const float* bias, float* emb_out, float* sum_out, float* bi_out, float* lr_out, float* fm_out, int B,
                   int F, int d, int* err_flag, int resume_passes, rs_stream_t stream);
/* Row gradient of every touched row = sum over its entries r (sample b = r / Fdiv) of
 *   gout[r,:] + dfm[b]*(S[b,:] - emb[r,:])      and, for the 1-dim tables, g1src[r] + dlr[b]
 * (each term optional).  Also writes rs_sparse_bwd_num_partials(F,cap,d) partial sums of ||g||^2.  ld_gout / ld_g1:
 * floats between consecutive rows of gout / values of g1src (0: packed, d and 1) -- the owner side of the row-sharded
 * exchange receives both in one buffer, the 1-dim gradient in a spare column behind the d-wide one. */
int rs_sparse_bwd_num_partials(int F, int cap, int d);
int rs_sparse_bwd_reduce(const int* plan, int F, int cap, int d, const float* gout, const float* g1src,
                         const float* dfm, const float* S, const float* emb, const float* dlr, int Fdiv,
                         float* chunk_sum, float* chunk_sum1, float* rowgrad, float* rowgrad1, float* norm_partials,
                         int ld_gout, int ld_g1, rs_stream_t stream);
/* rs_sparse_bwd_reduce that also writes, from extra workgroups of its second launch, n_dense_partials
 * sum-of-squares partials of a dense buffer (the flat gradient of the dense parameters): both are pieces of the one
 * global norm of clip_grad_norm_ (BaseModel.py:148).  cs_*: up to 8 column-sum finishes (the arguments of
 * rs_colsum_finish_multi; cs_count may be 0) that run beside its first launch -- the bias gradients of the dense
 * backward, which may lie inside `dense`; sk_*: likewise up to 8 split-K reductions (the arguments of
 * rs_splitk_reduce_multi) -- its weight gradients.  dense may be NULL with n_dense_partials = 0 (only the finishes
 * ride along). */
int rs_sparse_bwd_reduce_sumsq(const int* plan, int F, int cap, int d, const float* gout, const float* g1src,
                               const float* dfm, const float* S, const float* emb, const float* dlr, int Fdiv,
                               float* chunk_sum, float* chunk_sum1, float* rowgrad, float* rowgrad1, float* norm_partials,
                               const float* dense, long long dense_n, float* dense_partials, int n_dense_partials,
                               const float* const* cs_part, const int* cs_rows, const int* cs_n, const int* cs_ld,
                               float* const* cs_out, int cs_count, const float* const* sk_ws, float* const* sk_out,
                               const long long* sk_total, const int* sk_splitk, int sk_count, int ld_gout, int ld_g1,
                               rs_stream_t stream);
/* rs_sparse_bwd_reduce_sumsq whose row gradients go where a table of positions says: segment u of field f writes its
 * sum to position route[f * cap + u] of `rows` (ld_row floats per position) and its 1-dim sum to the same position of
 * `rows1` (ld_row1 floats per position; may be a column of `rows`); a negative position drops the gradient.  The
 * requester side of the row-sharded exchange: one gradient per distinct key straight into the bucket positions of the
 * send buffer (rs_shard_route's route), without a compact rowgrad and a packing launch in between. */
int rs_sparse_bwd_reduce_routed(const int* plan, int F, int cap, int d, const float* gout, const float* g1src,
                                const float* dfm, const float* S, const float* emb, const float* dlr, int Fdiv,
                                float* chunk_sum, float* chunk_sum1, float* rows, float* rows1, float* norm_partials,
                                const float* dense, long long dense_n, float* dense_partials, int n_dense_partials,
                                const float* const* cs_part, const int* cs_rows, const int* cs_n, const int* cs_ld,
                                float* const* cs_out, int cs_count, const float* const* sk_ws, float* const* sk_out,
                                const long long* sk_total, const int* sk_splitk, int sk_count, int ld_gout, int ld_g1,
                                const int* route, int ld_row, int ld_row1, rs_stream_t stream);
/* The same reduction for rows of W floats (W % 4 == 0, W <= 1024; field-aware FFM rows on the row-sharded path):
 * rowgrad[f, s, :] = sum of gout[sorted_src[q], :] over the entries of segment s, in ascending order; chunk_part
 * [F, cap, W] is scratch; norm_partials has rs_sparse_bwd_wide_num_partials(F, cap) floats.  The plan's window L
 * must be <= 64.  (BaseModel.py:141-150: the gradient of an nn.Embedding row is the sum over its occurrences.) */
int rs_sparse_bwd_wide_num_partials(int F, int cap);
int rs_sparse_bwd_reduce_wide(const int* plan, int F, int cap, int W, const float* gout, float* rowgrad, float* chunk_part,
                              float* norm_partials, rs_stream_t stream);
/* Optimizer step on the touched rows only (hyper: 8 floats, see rs_opt_prepare; scal[1] = clip coefficient).
 * last / last1 (int32 per row of table / table1, or null) with `step`: every stepped row is stamped with *step, the
 * number of optimizer steps it now contains (see rs_adam_catchup). */
int rs_sparse_update(const int* plan, int F, int cap, int d, int opt, const float* rowgrad, const float* rowgrad1,
                     float* table, float* m, float* v, float* table1, float* m1, float* v1, const float* hyper,
                     const float* scal, int* last, int* last1, const long long* step, rs_stream_t stream);
/* rs_sparse_update and rs_dense_update in one launch (optimizer.step() of BaseModel.py:149 over the touched table
 * rows and the flat dense parameter buffer). */
int rs_sparse_dense_update(const int* plan, int F, int cap, int d, int opt, const float* rowgrad, const float* rowgrad1,
                           float* table, float* m, float* v, float* table1, float* m1, float* v1, float* dense_w,
                           const float* dense_g, float* dense_m, float* dense_v, long long dense_n, const float* hyper,
                           const float* scal, int* last, int* last1, const long long* step, rs_stream_t stream);
/* Expand row gradients into dense [R,d] / [R] gradients (pre-zeroed by the caller): what nn.Embedding's
 * autograd produces for torch's dense optimizers. */
int rs_sparse_scatter_dense(const int* plan, int F, int cap, int d, const float* rowgrad, const float* rowgrad1,
                            float* grad, float* grad1, rs_stream_t stream);
/* Dense flat-buffer optimizer step, sums of squares, and nn.utils.clip_grad_norm_ (BaseModel.py:148):
 * scal[0] = sqrt(sum partials), scal[1] = min(1, max_norm/(scal[0]+1e-6)).
 * hyper = {lr, beta1, beta2, eps, weight_decay, lr/(1-beta1^t), sqrt(1-beta2^t), reserved}. */
int rs_dense_update(float* w, const float* g, float* m, float* v, long long n, int opt, const float* hyper,
                    const float* scal, rs_stream_t stream);
int rs_sumsq_partials(const float* x, long long n, float* partials, int nblocks, rs_stream_t stream);
int rs_clip_coef(const float* partials, int n, float max_norm, float* scal, rs_stream_t stream);
int rs_opt_prepare(long long* step, float* hyper, rs_stream_t stream);
int rs_clip_and_prepare(const float* partials, int n, float max_norm, float* scal, long long* step, float* hyper,
                        rs_stream_t stream);

/* ---- SGD with emb_regular on tables that are stepped on touched rows only -------------------------------------
 * BaseModel.regularization_loss (model/utils/BaseModel.py:214-231) adds lambda/2 ||W||^2 of every nn.Embedding to the
 * loss; under torch.optim.SGD (BaseModel.py:64-72) behind clip_grad_norm_ (:148) every row then has the gradient
 * g_r + lambda w_r, counts in the clipped norm, and rows the batch does not touch shrink by s = 1 - lr c lambda.
 * The tables are kept as w / G with one scalar G for all of them, so that shrink is G' = G s and no row is visited:
 *   rs_sparse_touched_dot: partials [F * rs_sparse_touched_dot_groups(cap, d)] = per-workgroup sums of
 *     <rowgrad[f, u, :], table[seg_row[f, u], :]> (+ rowgrad1 * table1) over the distinct rows of the plan;
 *   rs_clip_and_prepare_l2 = rs_clip_and_prepare with
 *     norm^2 = sum partials[n_sparse, n) + A + 2 lambda B + lambda^2 S,   A = sum partials[0, n_sparse) (the touched
 *     rows' ||g_r||^2), B = G * sum dot_partials, state = {G, S = sum ||w||^2 over every row of every table} (double);
 *     it leaves scal = {norm, c}, scal_tab = {norm, c / G'} (rs_sparse_update with SGD on the stored rows then is the
 *     reference's step w' = s w - lr c g), gscale = (float)G' for rs_gather_lr_fm_fwd_scaled, state = {G', S'} with
 *     S' = s^2 S - 2 s lr c B + lr^2 c^2 A, and adds lambda/2 S to *loss (optional);
 *   rs_scale_inplace: x *= (float)*scale -- turns stored rows back into weights (before a checkpoint / evaluation). */
int rs_sparse_touched_dot_groups(int cap, int d);
int rs_sparse_touched_dot(const int* plan, int F, int cap, int d, const float* rowgrad, const float* rowgrad1,
                          const float* table, const float* table1, float* partials, int n_partials, rs_stream_t stream);
int rs_clip_and_prepare_l2(const float* partials, int n_sparse, int n, const float* dot_partials, int n_dot,
                           float max_norm, float emb_l2, double* state, float* scal, float* scal_tab, float* gscale,
                           float* loss, long long* step, float* hyper, rs_stream_t stream);
int rs_scale_inplace(float* x, long long n, const double* scale, rs_stream_t stream);

/* ---- dense Adam / AdamW semantics on tables that are stepped on touched rows only ---------------------------
 * torch.optim.Adam / AdamW as the reference builds them (model/utils/BaseModel.py:64-72) step EVERY row of every
 * nn.Embedding table on every batch (model/utils/Layers.py:132: dense gradients): a row absent from a batch gets a
 * zero gradient, so its moments decay, AdamW shrinks it, and it keeps moving while its first moment is not zero.
 * rs_sparse_update skips those rows; these entry points replay the skipped zero-gradient steps when a row is next
 * read, so that lookups and updates see exactly the weights the reference would hold.
 *   last [R] int32 (one per table, zero-initialised): optimizer steps already contained in (table, m, v) of a row.
 *   step: the device step counter rs_clip_and_prepare increments (its value = steps taken so far).
 *   bctab [tab_n] float2 from rs_adam_bc_table: {1/(1-beta1^k), 1/sqrt(1-beta2^k)} in double precision as torch's
 *   host code computes them; steps beyond tab_n are computed on the fly.  kmax bounds the replay loop (the drift
 *   terms decay like beta1^j; ceil(log(1e-12)/log(beta1)) is enough), the rest is applied in closed form.
 * rs_adam_catchup: for every entry of the columnar batch idx [F,B], bring row row_offset[f]+idx[f,b] of table
 *   (width d) and/or table1 (width 1) up to `*step` steps and mark it as containing *step + mark_ahead steps.
 *   Call it before the lookup.  mark_ahead = 0 when the update that follows stamps the rows itself (rs_sparse_update
 *   with last / last1) and for forward-only passes; 1 when a training step's update cannot stamp (every entry then
 *   costs an atomic, against a plain load for rows that are up to date).  The learning rate must not have changed since a row's last
 *   step: call rs_adam_flush before changing hyper[0].
 * rs_adam_flush: every row of one table up to *step (before evaluation, state_dict, a learning-rate change). */
int rs_adam_bc_table(float* bctab, int tab_n, const float* hyper, rs_stream_t stream);
int rs_adam_catchup(const int64_t* idx, const int64_t* row_offset, const int64_t* vocab, int B, int F, int d,
                    float* table, float* m, float* v, int* last, float* table1, float* m1, float* v1, int* last1,
                    int opt, const long long* step, int mark_ahead, const float* hyper, const float* bctab, int tab_n,
                    int kmax, rs_stream_t stream);
/* rs_adam_catchup for keys that arrive as lists instead of a columnar batch: nlists = nsrc * F lists, list s * F + f of
 * field f starting at keys[sub_off[l]] with sub_cnt[l] <= per_list keys (the sub-list descriptors of
 * rs_sparse_bwd_sort / rs_sparse_bwd_merge_lists): the buckets a row-sharded owner receives before it serves them, or
 * the all-gathered distinct rows of every rank before a replicated table is stepped on their union.  No reference
 * counterpart (the reference is single-device). */
int rs_adam_catchup_lists(const int64_t* keys, const int* sub_off, const int* sub_cnt, int nlists, int per_list,
                          const int64_t* row_offset, const int64_t* vocab, int F, int d, float* table, float* m, float* v,
                          int* last, float* table1, float* m1, float* v1, int* last1, int opt, const long long* step,
                          int mark_ahead, const float* hyper, const float* bctab, int tab_n, int kmax, rs_stream_t stream);
/* rs_sparse_bwd_sort of a columnar batch and rs_adam_catchup (mark_ahead = 0) of the same batch in one launch: the sort
 * keeps one workgroup per field busy, the replay runs on the workgroups behind them.
 * split_passes > 0 (B <= rs_sort_split_max_entries()): only the first split_passes radix passes of the sort run here
 * and the plan is not written yet; rs_sort_gather(..., resume_passes = split_passes) finishes it beside the gather, so
 * that neither of the two launches in front of the first GEMM is as long as the whole sort. */
int rs_sort_split_max_entries(void);
int rs_sort_catchup(const int64_t* idx, const int* sub_off, const int* sub_cnt, const int* pay_base, int cap, int L,
                    const int64_t* vocab, const int64_t* row_offset, int* plan, int B, int F, int d, float* table, float* m,
                    float* v, int* last, float* table1, float* m1, float* v1, int* last1, int opt, const long long* step,
                    const float* hyper, const float* bctab, int tab_n, int kmax, int split_passes, rs_stream_t stream);
int rs_adam_flush(float* table, float* m, float* v, int* last, long long rows, int d, int opt, const long long* step,
                  const float* hyper, const float* bctab, int tab_n, int kmax, rs_stream_t stream);
/* The replay for the DISTINCT rows of a sorted batch (`plan` of rs_sparse_bwd_sort, F fields of capacity cap) -- no
 * atomics, a plan names a row once -- with the rows promised to the update that steps them next:
 *   mode 0: every row of the plan whose stamp is below *step is brought up to *step steps, and all rows of the plan are
 *           stamped *step + 1; the update that follows is then given no stamps to write (last = last1 = NULL).
 *   mode 1: the promise is taken back (stamps above *step return to *step) when that update does not come.
 *   mode 2: no promise -- rs_adam_catchup (mark_ahead = 0) once per distinct row instead of once per entry: rows below
 *           *step are brought up to it and stamped *step (the owner side of the row-sharded exchange, whose merged plan
 *           of the received key lists exists before the rows are served).
 * rs_sparse_dense_update_ahead is rs_sparse_dense_update (without stamps) with mode 0 for `next_plan`, the sorted plan
 * of the batch the NEXT step trains on, in the same launch: the rows `plan` steps carry the stamp *step from the
 * promise made to this update, so the replay leaves them alone, and every other row of the next batch is not touched
 * by the update.  What the reference's dense optimizer does to rows a batch does not touch (BaseModel.py:149) thus
 * costs a step no launch in front of its lookup. */
int rs_adam_catchup_plan(const int* plan, int F, int cap, int d, float* table, float* m, float* v, int* last, float* table1,
                         float* m1, float* v1, int* last1, int opt, const long long* step, const float* hyper,
                         const float* bctab, int tab_n, int kmax, int mode, rs_stream_t stream);
/* rs_adam_catchup_plan(mode 2) that also serves the rows: every row of the plan, once up to date, is written to each
 * position that names it (sorted_src of the entries of its segment) of `pay` (ld_pay floats per position; the d-wide row
 * at column col, the 1-dim value at column col1) -- rs_shard_gather for the owner of a row-sharded table whose rows the
 * replay has just had in its registers.  Rows of up to 256 floats. */
int rs_adam_catchup_serve(const int* plan, int F, int cap, int d, float* table, float* m, float* v, int* last, float* table1,
                          float* m1, float* v1, int* last1, int opt, const long long* step, const float* hyper,
                          const float* bctab, int tab_n, int kmax, float* pay, int ld_pay, int col, int col1,
                          rs_stream_t stream);
int rs_sparse_dense_update_ahead(const int* plan, int F, int cap, int d, int opt, const float* rowgrad, const float* rowgrad1,
                                 float* table, float* m, float* v, float* table1, float* m1, float* v1, float* dense_w,
                                 const float* dense_g, float* dense_m, float* dense_v, long long dense_n, const float* hyper,
                                 const float* scal, const int* next_plan, int* last, int* last1, const long long* step,
                                 const float* bctab, int tab_n, int kmax, rs_stream_t stream);


/* ---- row-sharded tables: device side of the all-to-all exchange (no reference counterpart: the
 * reference is single-device; semantics are those of Embedding_layer.forward / its backward on the
 * union of all ranks' batches).  Row i of a field lives on rank i % N at local index i / N + 1. ---- */
/* Buckets: (owner o, field f) holds up to capf[f] distinct keys (device int32 [F]; a field with few rows needs no more
 * than its rows per owner); boff[f] = capf[0] + .. + capf[f-1], S = boff[F].  Key buffers are int64 [N, S + F] with
 * bucket (o, f) at o*(S+F) + boff[f] + f (count slot, then the keys); row and gradient buffers hold ld >= d floats per
 * position, row r of bucket (o, f) at position o*S + boff[f] + r, which is also what route[f][u] holds for the u-th
 * distinct key of field f (-1: the key did not fit, *overflow_flag is set).  With ld > d the value (or gradient) of the
 * 1-dim LR table travels in a spare column of the same buffer: one all-to-all each way instead of two.
 * rs_shard_counts: the owner's sub-list lengths from the count slots of the received buckets and, in the same launch,
 * slot_map (int32 [F, B], may be NULL): per entry (f, b) of the requester's batch (its sorted `plan` of capacity cap and
 * `route`) the position of its key's row, N*S -- a row of zeros the caller keeps behind the received rows -- for
 * padding entries and lost keys; Fdiv: sample of gradient row r = r / Fdiv. */
int rs_shard_route(const int* plan, int F, int cap, const int64_t* row_offset, int N, const int* capf, const int* boff,
                   int S, long long* send_idx, int* route, int* overflow_flag, rs_stream_t stream);
int rs_shard_gather(const long long* recv_idx, const int64_t* local_row_offset, const int64_t* local_vocab, int N,
                    int F, const int* capf, const int* boff, int S, int cap_max, int d, const float* table,
                    const float* table1, float* rows, float* rows1, int ld, int ld1, int* err_flag, rs_stream_t stream);
/* The requester's lookup in the received rows: rs_gather_lr_fm_fwd (Layers.py:137-160, :107-109, FM.py:27-30; same
 * outputs, same arithmetic and summation order) with entry (f, b) reading position slot_map[f*B + b] of `rows` (ld
 * floats per position) and of `rows1` (the 1-dim values, ld1 floats apart) instead of a table row. */
int rs_shard_lookup(const int* slot_map, const float* rows, int ld, const float* rows1, int ld1, const float* bias,
                    float* emb_out, float* sum_out, float* bi_out, float* lr_out, float* fm_out, int B, int F, int d,
                    rs_stream_t stream);
/* The plain expansion (every entry copies the row of its distinct key), for rows wider than rs_shard_lookup takes. */
int rs_shard_expand(const int* plan, const int* route, int F, int cap, int d, const float* rows,
                    const float* rows1, float* emb, float* lr1, rs_stream_t stream);
int rs_shard_pack(const int* plan, const int* route, int F, int cap, int d, const float* rowgrad,
                  const float* rowgrad1, float* grads, float* grads1, int ld, int ld1, rs_stream_t stream);
int rs_shard_counts(const long long* recv_idx, int N, int F, const int* capf, const int* boff, int S, int* sub_cnt,
                    const int* plan, const int* route, int cap, int* slot_map, int B, int Fdiv, rs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RS_HIP_H */
