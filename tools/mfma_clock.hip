// Sustained shader clock and fp32 MFMA rate under a pure-MFMA load: every wave runs a chain of
// v_mfma_f32_32x32x2_f32 and reads the shader clock (clock64) and the 100 MHz wall clock (wall_clock64).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, long long* t, int iters, int chains) {
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (chains == 1) {
#pragma unroll
            for (int u = 0; u < 16; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = w1 - w0; }
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k16(float* out, long long* t, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int r = 0; r < 4; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main() {
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu, iters = 400;
        float* out; long long* t;
        (void)hipMalloc(&out, grid * 256 * 4); (void)hipMalloc(&t, grid * 16);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k16<<<grid, 256>>>(out, t, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k16<<<grid, 256>>>(out, t, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        const double flop = (double)grid * 4 * iters * 32 * 2048.0;
        printf("16x16x4 f32, 4 chains  wgs/CU %d  %.1f us  %.1f TFLOP/s  shader clock %.0f MHz  cycles/MFMA/wave %.1f\n", wgs_per_cu,
               ms * 1e3, flop / ms / 1e9, 100.0 * h[0] / h[1], (double)h[0] / (iters * 32));
        (void)hipFree(out); (void)hipFree(t);
    }
    const int iters = 400;
    for (int chains = 1; chains <= 4; chains += 3)
        for (int wgs_per_cu = 1; wgs_per_cu <= 4; ++wgs_per_cu) {
            const int grid = 256 * wgs_per_cu;
            float* out; long long* t;
            hipMalloc(&out, grid * 256 * 4); hipMalloc(&t, grid * 16);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<grid, 256>>>(out, t, 10, chains);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<grid, 256>>>(out, t, iters, chains);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            const double flop = (double)grid * 4 * iters * 16 * 4096.0;
            printf("chains %d  wgs/CU %d  %.1f us  %.1f TFLOP/s  shader clock %.0f MHz (cycles %lld, wall ticks %lld)\n", chains,
                   wgs_per_cu, ms * 1e3, flop / ms / 1e9, 100.0 * h[0] / h[1], h[0], h[1]);
            hipFree(out); hipFree(t);
        }
    return 0;
}
