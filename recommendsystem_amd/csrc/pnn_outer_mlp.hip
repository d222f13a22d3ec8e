// The first MLP layer of PNN over [x_other | outer products of all field pairs] (model/PNN/PNN.py:19-27,58,80-103)
// without the (B, P*d*d) matrix of products: 189,696 columns at F = 39, d = 16 -- 3.1 GB at B = 4096, 12.4 GB at 16384.
//
// The reduction dimension is walked a block of pairs at a time (pairs are numbered first field major, so a block is
// a range [i_lo, i_hi) of first fields): the block's products are written to a buffer sized to stay in the Infinity
// Cache, multiplied into the running sum by the matrix-core GEMM (rs_gemm_f32_acc), and the buffer is reused.  The
// backward recomputes each block to form its columns of dW and its share of demb.  The weight's rows hold
// K0 + P*d*d floats -- not a multiple of four -- so the weight is first copied into a shadow whose two column groups
// start 16-byte aligned (every product then runs on the LDS-DMA GEMM); dW is built in that layout and copied back.
// Host code only: the kernels are those of rs_pnn_outer_{fwd,bwd}_range, rs_gemm_f32(_acc) and rs_colsum_f32.
#include "common.h"
#include "../../include/rs_hip.h"

namespace {

constexpr long long OUTER_BLOCK_BYTES = 192ll << 20;    // one block of products: inside the 256 MB Infinity Cache
constexpr int MAX_BLOCKS = 64;

__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, long long lds, float* __restrict__ dst,
                                                        long long ldd, int rows, int cols, int cols_padded) {
    // dst[r, c] = src[r, c] for c < cols, 0 for cols <= c < cols_padded; blockIdx.y walks the rows
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const float* s = src + (long long)r * lds;
        float* d = dst + (long long)r * ldd;
        for (int c = blockIdx.x * 256 + threadIdx.x; c < cols_padded; c += gridDim.x * 256) d[c] = c < cols ? s[c] : 0.f;
    }
}

void copy_cols(const float* src, long long lds, float* dst, long long ldd, int rows, int cols, int cols_padded,
               hipStream_t stream) {
    if (rows <= 0 || cols_padded <= 0) return;
    const int gx = (cols_padded + 255) / 256 < 256 ? (cols_padded + 255) / 256 : 256;
    const int gy = rows < 4096 ? rows : 4096;
    hipLaunchKernelGGL(copy_cols_kernel, dim3(gx, gy), dim3(256), 0, stream, src, lds, dst, ldd, rows, cols, cols_padded);
}

inline long long pair_start(int i, int F) { return (long long)i * (2 * F - i - 1) / 2; }

struct Shape {
    int B, F, d, K0, N;
    int K0p;                   // K0 rounded up to four floats
    long long KO, Ksh;         // P*d*d; row length of the aligned shadow
    int nblocks, lo[MAX_BLOCKS], hi[MAX_BLOCKS];
    long long wmax;            // widest block, in columns
    size_t gemm_ws;            // floats: the largest split-K workspace of any product
};

size_t gemm_ws_floats(int M, int N, int K) {
    int sk = 1;
    rs_gemm_plan(M, N, K, 256, &sk);
    return sk > 1 ? rs_gemm_workspace_bytes(M, N, sk) / sizeof(float) : 0;
}

bool make_shape(Shape& s, int B, int F, int d, int K0, int N, long long block_bytes) {
    s.B = B; s.F = F; s.d = d; s.K0 = K0; s.N = N;
    s.K0p = (K0 + 3) / 4 * 4;
    const long long dd = (long long)d * d;
    s.KO = pair_start(F - 1, F) * dd;
    s.Ksh = s.K0p + s.KO;
    long long budget = (block_bytes > 0 ? block_bytes : OUTER_BLOCK_BYTES) / ((long long)B * dd * 4);
    if (budget < 1) budget = 1;
    s.nblocks = 0;
    s.wmax = 0;
    s.gemm_ws = 0;
    int lo = 0;
    while (lo < F - 1) {
        int hi = lo + 1;
        long long n = F - 1 - lo;
        while (hi < F - 1 && n + (F - 1 - hi) <= budget) { n += F - 1 - hi; ++hi; }
        if (s.nblocks == MAX_BLOCKS) return false;
        s.lo[s.nblocks] = lo; s.hi[s.nblocks] = hi; ++s.nblocks;
        const long long w = n * dd;
        if (w > s.wmax) s.wmax = w;
        if (w > 0x7fffffffll) return false;
        size_t g = gemm_ws_floats(B, N, (int)w);                  // z += Xb Wb^T
        const size_t g2 = gemm_ws_floats(N, (int)w, B);           // dWb = dz^T Xb
        const size_t g3 = gemm_ws_floats(B, (int)w, N);           // G = dz Wb
        g = g > g2 ? g : g2;
        g = g > g3 ? g : g3;
        if (g > s.gemm_ws) s.gemm_ws = g;
        lo = hi;
    }
    if (K0 > 0) {
        size_t g = gemm_ws_floats(B, N, s.K0p);
        const size_t g2 = gemm_ws_floats(N, s.K0p, B), g3 = gemm_ws_floats(B, s.K0p, N);
        g = g > g2 ? g : g2;
        g = g > g3 ? g : g3;
        if (g > s.gemm_ws) s.gemm_ws = g;
    }
    return true;
}

inline size_t up64(size_t n) { return (n + 63) & ~(size_t)63; }       // pieces start 256-byte aligned

// carving of the caller's workspace
struct Carve {
    float *Ws, *xo, *buf, *gemm, *dWs, *gbuf, *dxo, *colsum;
    size_t total;
};

Carve carve(const Shape& s, float* ws, bool backward) {
    Carve c{};
    size_t off = 0;
    auto take = [&](size_t n) { float* p = ws ? ws + off : nullptr; off += up64(n); return p; };
    c.Ws = take((size_t)s.N * s.Ksh);
    c.xo = take((size_t)s.B * s.K0p);
    c.buf = take((size_t)s.B * s.wmax);
    c.gemm = take(s.gemm_ws);
    if (backward) {
        c.dWs = take((size_t)s.N * s.Ksh);
        c.gbuf = take((size_t)s.B * s.wmax);
        c.dxo = take((size_t)s.B * s.K0p);
        c.colsum = take(rs_colsum_workspace_bytes(s.B, s.N) / sizeof(float));
    }
    c.total = off;
    return c;
}

// the weight [N, K0 + KO] into its aligned shadow [N, K0p + KO]; x_other [B, K0] (row stride ldx) into [B, K0p]
void stage_operands(const Shape& s, const Carve& c, const float* W, const float* x_other, long long ldx, hipStream_t stream) {
    const long long ldw = (long long)s.K0 + s.KO;
    if (s.K0 > 0) {
        copy_cols(W, ldw, c.Ws, s.Ksh, s.N, s.K0, s.K0p, stream);
        copy_cols(x_other, ldx, c.xo, s.K0p, s.B, s.K0, s.K0p, stream);
    }
    // the outer-product columns, in pieces of at most 2^30 columns per row (int arguments)
    for (long long done = 0; done < s.KO; done += 1 << 30) {
        const int w = (int)(s.KO - done < (1 << 30) ? s.KO - done : (1 << 30));
        copy_cols(W + s.K0 + done, ldw, c.Ws + s.K0p + done, s.Ksh, s.N, w, w, stream);
    }
}

}  // namespace

extern "C" {

size_t rs_pnn_outer_mlp_workspace_floats(int B, int F, int d, int K0, int N, int backward, long long block_bytes) {
    Shape s;
    if (B <= 0 || F < 2 || d <= 0 || K0 < 0 || N <= 0 || !make_shape(s, B, F, d, K0, N, block_bytes)) return 0;
    return carve(s, nullptr, backward != 0).total;
}

// Number of blocks of pairs the two calls below walk (0: unsupported shape).
int rs_pnn_outer_mlp_blocks(int B, int F, int d, long long block_bytes) {
    Shape s;
    if (B <= 0 || F < 2 || d <= 0 || !make_shape(s, B, F, d, 0, 1, block_bytes)) return 0;
    return s.nblocks;
}

int rs_pnn_outer_mlp_fwd(const float* emb, const float* x_other, long long ldx, int B, int F, int d, int K0, const float* W,
                         const float* bias, int N, float* z, float* workspace, size_t workspace_floats, long long block_bytes,
                         hipStream_t stream) {
    RS_CHECK_ARG(emb && W && z && workspace, "rs_pnn_outer_mlp_fwd: null argument");
    RS_CHECK_ARG(B > 0 && F >= 2 && d > 0 && K0 >= 0 && N > 0, "rs_pnn_outer_mlp_fwd: bad sizes");
    RS_CHECK_ARG(K0 == 0 || (x_other && ldx >= K0), "rs_pnn_outer_mlp_fwd: x_other missing or its row stride too small");
    RS_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "rs_pnn_outer_mlp_fwd: workspace must be 16-byte aligned");
    Shape s;
    RS_CHECK_ARG(make_shape(s, B, F, d, K0, N, block_bytes), "rs_pnn_outer_mlp_fwd: F=%d d=%d B=%d needs more than %d blocks of pairs", F, d, B, MAX_BLOCKS);
    const Carve c = carve(s, workspace, false);
    if (workspace_floats < c.total) {
        rs_set_error("rs_pnn_outer_mlp_fwd: workspace of %zu floats, %zu needed", workspace_floats, c.total);
        return RS_ERR_WORKSPACE;
    }
    stage_operands(s, c, W, x_other, ldx, stream);
    RS_CHECK_LAUNCH("rs_pnn_outer_mlp_fwd");
    const size_t gbytes = s.gemm_ws * sizeof(float);
    bool started = false;
    int rc;
    if (K0 > 0) {
        rc = rs_gemm_f32(0, 0, B, N, s.K0p, c.xo, s.K0p, c.Ws, (int)s.Ksh, z, N, nullptr, 0, nullptr, 0, 0, -1, 0, c.gemm, gbytes,
                         nullptr, stream);
        if (rc != RS_OK) return rc;
        started = true;
    }
    const long long dd = (long long)d * d;
    for (int k = 0; k < s.nblocks; ++k) {
        const long long off = s.K0p + pair_start(s.lo[k], F) * dd;
        const int w = (int)((pair_start(s.hi[k], F) - pair_start(s.lo[k], F)) * dd);
        rc = rs_pnn_outer_fwd_range(emb, B, F, d, s.lo[k], s.hi[k], c.buf, w, stream);
        if (rc != RS_OK) return rc;
        const float* bk = k == s.nblocks - 1 ? bias : nullptr;
        if (started) rc = rs_gemm_f32_acc(0, 0, B, N, w, c.buf, w, c.Ws + off, (int)s.Ksh, z, N, bk, 0, -1, 0, c.gemm, gbytes, stream);
        else rc = rs_gemm_f32(0, 0, B, N, w, c.buf, w, c.Ws + off, (int)s.Ksh, z, N, bk, 0, nullptr, 0, 0, -1, 0, c.gemm, gbytes, nullptr, stream);
        if (rc != RS_OK) return rc;
        started = true;
    }
    return RS_OK;
}

int rs_pnn_outer_mlp_bwd(const float* emb, const float* x_other, long long ldx, const float* dz, int B, int F, int d, int K0,
                         const float* W, int N, float* dW, float* db, float* demb, float* dx_other, float* workspace,
                         size_t workspace_floats, long long block_bytes, hipStream_t stream) {
    RS_CHECK_ARG(emb && W && dz && dW && demb && workspace, "rs_pnn_outer_mlp_bwd: null argument");
    RS_CHECK_ARG(B > 0 && F >= 2 && d > 0 && K0 >= 0 && N > 0, "rs_pnn_outer_mlp_bwd: bad sizes");
    RS_CHECK_ARG(K0 == 0 || (x_other && ldx >= K0), "rs_pnn_outer_mlp_bwd: x_other missing or its row stride too small");
    RS_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "rs_pnn_outer_mlp_bwd: workspace must be 16-byte aligned");
    Shape s;
    RS_CHECK_ARG(make_shape(s, B, F, d, K0, N, block_bytes), "rs_pnn_outer_mlp_bwd: F=%d d=%d B=%d needs more than %d blocks of pairs", F, d, B, MAX_BLOCKS);
    const Carve c = carve(s, workspace, true);
    if (workspace_floats < c.total) {
        rs_set_error("rs_pnn_outer_mlp_bwd: workspace of %zu floats, %zu needed", workspace_floats, c.total);
        return RS_ERR_WORKSPACE;
    }
    stage_operands(s, c, W, x_other, ldx, stream);
    RS_CHECK_LAUNCH("rs_pnn_outer_mlp_bwd");
    const size_t gbytes = s.gemm_ws * sizeof(float);
    const int Ksh = (int)s.Ksh;
    int rc;
    if (K0 > 0) {
        // dW[:, :K0] = dz^T x_other;  dx_other = dz W[:, :K0]
        rc = rs_gemm_f32(1, 1, N, s.K0p, B, dz, N, c.xo, s.K0p, c.dWs, Ksh, nullptr, 0, nullptr, 0, 0, -1, 0, c.gemm, gbytes, nullptr, stream);
        if (rc != RS_OK) return rc;
        if (dx_other) {
            rc = rs_gemm_f32(0, 1, B, s.K0p, N, dz, N, c.Ws, Ksh, c.dxo, s.K0p, nullptr, 0, nullptr, 0, 0, -1, 0, c.gemm, gbytes, nullptr, stream);
            if (rc != RS_OK) return rc;
            copy_cols(c.dxo, s.K0p, dx_other, K0, B, K0, K0, stream);
        }
    }
    const long long dd = (long long)d * d;
    for (int k = 0; k < s.nblocks; ++k) {
        const long long off = s.K0p + pair_start(s.lo[k], F) * dd;
        const int w = (int)((pair_start(s.hi[k], F) - pair_start(s.lo[k], F)) * dd);
        rc = rs_pnn_outer_fwd_range(emb, B, F, d, s.lo[k], s.hi[k], c.buf, w, stream);
        if (rc != RS_OK) return rc;
        rc = rs_gemm_f32(1, 1, N, w, B, dz, N, c.buf, w, c.dWs + off, Ksh, nullptr, 0, nullptr, 0, 0, -1, 0, c.gemm, gbytes, nullptr, stream);
        if (rc != RS_OK) return rc;
        rc = rs_gemm_f32(0, 1, B, w, N, dz, N, c.Ws + off, Ksh, c.gbuf, w, nullptr, 0, nullptr, 0, 0, -1, 0, c.gemm, gbytes, nullptr, stream);
        if (rc != RS_OK) return rc;
        rc = rs_pnn_outer_bwd_range(emb, c.gbuf, w, B, F, d, s.lo[k], s.hi[k], k > 0, demb, stream);
        if (rc != RS_OK) return rc;
    }
    // dW back from the shadow layout
    const long long ldw = (long long)K0 + s.KO;
    if (K0 > 0) copy_cols(c.dWs, s.Ksh, dW, ldw, N, K0, K0, stream);
    for (long long done = 0; done < s.KO; done += 1 << 30) {
        const int w = (int)(s.KO - done < (1 << 30) ? s.KO - done : (1 << 30));
        copy_cols(c.dWs + s.K0p + done, s.Ksh, dW + K0 + done, ldw, N, w, w, stream);
    }
    RS_CHECK_LAUNCH("rs_pnn_outer_mlp_bwd");
    if (db) {
        rc = rs_colsum_f32(dz, B, N, N, nullptr, db, c.colsum, rs_colsum_workspace_bytes(B, N), stream);
        if (rc != RS_OK) return rc;
    }
    return RS_OK;
}

}  // extern "C"
