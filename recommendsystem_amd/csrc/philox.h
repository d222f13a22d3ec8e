// Counter-based random bits for dropout (nn.Dropout in training mode, model/utils/Layers.py:33-37,46).
// keep(i) of element i of a tensor depends only on (seed, salt, call, i): the backward pass, or the epilogue of the GEMM
// that produces or consumes the tensor, regenerates the forward's mask from the same four numbers instead of reading a
// stored one.  One Philox4x32-10 block serves the four consecutive elements 4*blk .. 4*blk+3.
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// What a kernel needs to regenerate a mask (kernel-argument form; p <= 0 or off: no mask).
struct DropKey {
    float p;
    unsigned salt;
    unsigned long long seed;
    unsigned long long call_host;
    const long long* call_dev;     // device counter added to call_host (a replayed graph draws fresh masks), or null
};

struct DropConsts {
    unsigned thr;                  // keep iff u >= thr (u uniform on 32 bits): P(keep) = 1 - p
    float scale;                   // 1 / (1 - p), 0 when everything is dropped
    bool drop_all;
    unsigned k0, k1, c2, c3;
};

__device__ __forceinline__ DropConsts drop_consts(const DropKey& k) {
    DropConsts d;
    const unsigned long long call = k.call_host + (k.call_dev ? (unsigned long long)*k.call_dev : 0ull);
    const double thr_d = (double)k.p * 4294967296.0;
    d.thr = thr_d >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)thr_d;
    d.drop_all = k.p >= 1.f;
    d.scale = d.drop_all ? 0.f : 1.f / (1.f - k.p);
    d.k0 = (unsigned)k.seed; d.k1 = (unsigned)(k.seed >> 32);
    d.c2 = (unsigned)call; d.c3 = k.salt ^ (unsigned)(call >> 32);
    return d;
}

// multipliers (0 or 1/(1-p)) of the elements 4*blk .. 4*blk+3
__device__ __forceinline__ void drop_factors(const DropConsts& d, long long blk, float (&f)[4]) {
    unsigned c[4] = {(unsigned)blk, (unsigned)((unsigned long long)blk >> 32), d.c2, d.c3};
    philox4x32_10(c, d.k0, d.k1);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = (!d.drop_all && c[e] >= d.thr) ? d.scale : 0.f;
}

}  // namespace
