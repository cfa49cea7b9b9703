// Probe: sustained bf16 MFMA rate of the whole chip with operands in registers (no memory traffic): 16x16x32 and 32x32x16,
// 1, 2 or 4 waves per SIMD, independent accumulators.  Reports TFLOP/s and the implied clock if an MFMA 16x16x32 takes 16
// cycles (32x32x16: 32) -- what the nominal 2.5 PFLOP/s assumes at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x + 2 * e)); }
    float r = 0.f;
    if (KIND == 0) {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) r += c[i][0];
    } else {
        f32x16 c[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) r += c[i][0];
    }
    if (r == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
void run(const char* name, int wg_per_cu, float* out) {
    const int iters = 20000, grid = 256 * wg_per_cu;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_mfma<KIND>), dim3(grid), dim3(256), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k_mfma<KIND>), dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double per = KIND == 0 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2;
    const double n_mfma = (double)iters * (KIND == 0 ? 32 : 16) * 4 * grid;          // per wave x 4 waves x workgroups
    const double tf = n_mfma * per / (ms * 1e-3) / 1e12;
    const double cyc_per = KIND == 0 ? 16.0 : 32.0;
    const double clk = (double)iters * (KIND == 0 ? 32 : 16) * wg_per_cu * cyc_per / (ms * 1e-3) / 1e9;
    printf("%-14s %d wave(s) per SIMD: %8.2f ms, %7.1f TFLOP/s, implied clock %.2f GHz\n", name, wg_per_cu, ms, tf, clk);
}

int main() {
    float* out; (void)hipMalloc(&out, 1 << 22);
    for (int w : {1, 2, 4}) run<0>("16x16x32 bf16", w, out);
    for (int w : {1, 2, 4}) run<1>("32x32x16 bf16", w, out);
    return 0;
}
