// Store-path probe: 256 workgroups x 8 waves write a [65536 x 1024] bf16 matrix as 256 x 256 tiles (4 per workgroup), each
// wave its 128 x 64 block with sixteen 16-byte stores per lane, in two lane->address patterns:
//   0: lane (r16, g) -> row r16, bytes 16 g       (one instruction = 16 rows x 64 bytes: half lines; the GEMM epilogue)
//   1: lane L        -> row L / 8, bytes 16 (L % 8) (one instruction = 8 rows x 128 bytes: whole lines)
// build: hipcc --offload-arch=gfx950 -O3 -o tests/probes/bin/store_pattern tests/probes/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT>
__global__ __launch_bounds__(512) void k(u16* C, int ldc, int tiles_n, int ntiles) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 2, wc = w & 3;
    const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int m0 = (t / tiles_n) * 256 + wr * 128, n0 = (t % tiles_n) * 256 + wc * 64;
        if (PAT == 0) {
            const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
                    *(u32x4*)(C + (size_t)(m0 + 16 * mi + r16) * ldc + n0 + 32 * j + 8 * g) = v;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                *(u32x4*)(C + (size_t)(m0 + 8 * i + (lane >> 3)) * ldc + n0 + 8 * (lane & 7)) = v;
        }
    }
}

int main() {
    const int M = 65536, N = 1024;
    u16* C;
    hipMalloc(&C, (size_t)M * N * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int pat = 0; pat < 2; ++pat) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) {
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, C, N, N / 256, (M / 256) * (N / 256));
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, C, N, N / 256, (M / 256) * (N / 256));
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("pattern %d: %.1f us per pass, %.2f TB/s\n", pat, ms / 20 * 1e3, (double)M * N * 2 / (ms / 20 * 1e-3) / 1e12);
        }
    }
    return 0;
}
