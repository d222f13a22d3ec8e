// Shared helpers for the gfx950 kernels of librs_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define RS_OK 0
#define RS_ERR_ARG (-1)       // bad argument / unsupported shape
#define RS_ERR_LAUNCH (-2)    // hipLaunch / runtime error
#define RS_ERR_WORKSPACE (-3) // workspace too small

void rs_set_error(const char* fmt, ...);

#define RS_CHECK_ARG(cond, ...)                                                                    \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            rs_set_error(__VA_ARGS__);                                                             \
            return RS_ERR_ARG;                                                                     \
        }                                                                                          \
    } while (0)

#define RS_CHECK_LAUNCH(name)                                                                      \
    do {                                                                                           \
        hipError_t e__ = hipGetLastError();                                                        \
        if (e__ != hipSuccess) {                                                                   \
            rs_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));                   \
            return RS_ERR_LAUNCH;                                                                  \
        }                                                                                          \
    } while (0)

static inline int rs_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- device helpers -------------------------------------------------------------------------
#define RS_WAVE 64

__device__ __forceinline__ float rs_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a block of NT threads (NT multiple of 64, <= 1024); result valid in every thread.
template <int NT>
__device__ __forceinline__ float rs_block_sum(float v, float* smem /* >= NT/64 floats */) {
    v = rs_wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += smem[i];   // fixed order: deterministic
    return t;
}

// "Last workgroup finishes the reduction": every workgroup of a launch writes its partial results with
// rs_st_agent, then calls rs_last_arrival (all threads, uniformly); exactly one workgroup -- the last to
// arrive at `counter` -- gets true and may read every workgroup's partials with rs_ld_agent.  The partials
// travel as device-scope (sc1) stores and loads, so no cache-wide write-back or invalidate is needed: a
// workgroup only waits for its own stores before it signals.  The counter resets itself, so one
// zero-initialised int serves every launch issued in order on a stream.  s_flag: one int of LDS.
__device__ __forceinline__ void rs_st_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float rs_ld_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool rs_last_arrival(int* counter, int expected, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's partial stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = atomicAdd(counter, 1);
        const int last = old == expected - 1;
        if (last) atomicExch(counter, 0);
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0;
}

// Activation codes shared with the host side.
enum { RS_ACT_NONE = 0, RS_ACT_RELU = 1, RS_ACT_TANH = 2, RS_ACT_LEAKYRELU = 3, RS_ACT_PRELU = 4 };

__device__ __forceinline__ float rs_act(int act, float z, float slope) {
    switch (act) {
        case RS_ACT_RELU: return z > 0.f ? z : 0.f;
        case RS_ACT_TANH: return tanhf(z);
        case RS_ACT_LEAKYRELU: return z > 0.f ? z : 0.01f * z;
        case RS_ACT_PRELU: return z > 0.f ? z : slope * z;
        default: return z;
    }
}

// d act(z)/dz given z (pre-activation) and h = act(z).
__device__ __forceinline__ float rs_act_grad(int act, float z, float h, float slope) {
    switch (act) {
        case RS_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case RS_ACT_TANH: return 1.f - h * h;
        case RS_ACT_LEAKYRELU: return z > 0.f ? 1.f : 0.01f;
        case RS_ACT_PRELU: return z > 0.f ? 1.f : slope;
        default: return 1.f;
    }
}
