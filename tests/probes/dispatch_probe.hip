// Probe: cost of dispatching 2048 x 256-thread workgroups with N KB of LDS; VALU rate sanity.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDSB>
__global__ __launch_bounds__(256) void k_empty(float* out, int iters) {
    __shared__ float lds[LDSB / 4];
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (threadIdx.x == 0) lds[0] = v;
    __syncthreads();
    if (v == 12345.f) out[blockIdx.x] = lds[0];
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / reps;
}
int main() {
    float* d; hipMalloc(&d, 1 << 20);
    for (int grid : {256, 2048, 8192}) {
        printf("grid %5d: lds1K iters0 %.1f us | lds42K iters0 %.1f us | lds42K iters1000 %.1f us | lds1K iters1000 %.1f us | dim3(16,8,16) lds42K %.1f us\n", grid,
               timeit([&] { hipLaunchKernelGGL(k_empty<1024>, dim3(grid), dim3(256), 0, 0, d, 0); }, 20),
               timeit([&] { hipLaunchKernelGGL(k_empty<43008>, dim3(grid), dim3(256), 0, 0, d, 0); }, 20),
               timeit([&] { hipLaunchKernelGGL(k_empty<43008>, dim3(grid), dim3(256), 0, 0, d, 1000); }, 20),
               timeit([&] { hipLaunchKernelGGL(k_empty<1024>, dim3(grid), dim3(256), 0, 0, d, 1000); }, 20),
               timeit([&] { hipLaunchKernelGGL(k_empty<43008>, dim3(16, 8, grid / 128), dim3(256), 0, 0, d, 0); }, 20));
    }
    return 0;
}
