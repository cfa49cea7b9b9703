// Probe: cost of finishing a split-K weight-gradient GEMM with fp32 atomics vs fp32 slabs + a reduce pass.
// 256 workgroups x 512 threads; each holds a 256 x 256 fp32 tile (128 values per lane, MFMA C layout) and either
//   A) atomically adds it into out[N x K] (tiles overlap `slices`-fold), or
//   B) stores it to its own slab (16-byte stores), followed by a reduce kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(512) void finish(float* __restrict__ out, float* __restrict__ slabs, int N, int K, int tiles_k, int slices) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int wr = w >> 2, wc = w & 3;
    const int tile = blockIdx.x / slices, slice = blockIdx.x % slices;
    const int n0 = (tile / tiles_k) * 256, k0 = (tile % tiles_k) * 256;
    float* dst = MODE == 0 ? out : slabs + (size_t)slice * N * K;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int n = n0 + wr * 128 + 16 * mi + r16;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int k = k0 + wc * 64 + 16 * ni + 4 * g;
            f32x4 v = {1.f + mi, 2.f + ni, 3.f, (float)lane};
            float* p = dst + (size_t)n * K + k;
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(p + e), v[e]);
            } else {
                *(f32x4*)p = v;
            }
        }
    }
}
__global__ void reduce(float* __restrict__ out, const float* __restrict__ slabs, size_t n, int slices) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t step = (size_t)gridDim.x * blockDim.x * 4;
    for (; i < n; i += step) {
        f32x4 s = *(const f32x4*)(out + i);
        for (int k = 0; k < slices; ++k) s += *(const f32x4*)(slabs + (size_t)k * n + i);
        *(f32x4*)(out + i) = s;
    }
}
int main() {
    const int shapes[4][2] = {{1536, 512}, {512, 512}, {1024, 512}, {768, 512}};
    for (int s = 0; s < 4; ++s) {
        const int N = shapes[s][0], K = shapes[s][1];
        const int tiles_k = K / 256, tiles = (N / 256) * tiles_k, slices = 256 / tiles;
        float *out, *slabs;
        hipMalloc(&out, (size_t)N * K * 4);
        hipMalloc(&slabs, (size_t)slices * N * K * 4);
        hipMemset(out, 0, (size_t)N * K * 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms;
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < 20; ++it) {
                    if (mode == 0) hipLaunchKernelGGL(finish<0>, dim3(tiles * slices), dim3(512), 0, 0, out, slabs, N, K, tiles_k, slices);
                    else {
                        hipLaunchKernelGGL(finish<1>, dim3(tiles * slices), dim3(512), 0, 0, out, slabs, N, K, tiles_k, slices);
                        hipLaunchKernelGGL(reduce, dim3(1024), dim3(256), 0, 0, out, slabs, (size_t)N * K, slices);
                    }
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%dx%d slices %d  %s: %.1f us per GEMM finish\n", N, K, slices, mode == 0 ? "atomics" : "slabs+reduce", ms * 1000 / 20);
        }
        float h[4]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("   check out[0..3] = %g %g %g %g\n", h[0], h[1], h[2], h[3]);
        hipFree(out); hipFree(slabs);
    }
    return 0;
}
