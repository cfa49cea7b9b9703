// Probe: what limits a GEMM epilogue's stores?  256 workgroups x 512 threads each write 256 x 256 bf16 tiles of a
// [M x N] matrix (N = 1536, tile-strided like the GEMM) `reps` times, 16-byte stores, in three lane->address patterns:
//   0: MFMA-C pattern of gemm8 (per instruction: 16 rows x 64 contiguous bytes)
//   1: full lines (per instruction: 8 rows x 128 contiguous bytes)
//   2: 4 rows x 256 bytes     3: pattern 1 with non-temporal stores
// plus a VALU-only variant to see the issue floor.  Prints GB/s and cycles per tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT>
__global__ __launch_bounds__(512) void wr(unsigned short* __restrict__ C, int ldc, int tiles_n, int ntiles, int tiles_per_wg) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr_ = w >> 2, wc = w & 3, r16 = lane & 15, g = lane >> 4;
    for (int it = 0; it < tiles_per_wg; ++it) {
        const int tile = (blockIdx.x + it * gridDim.x) % ntiles;
        const int m0 = (tile / tiles_n) * 256 + wr_ * 128, n0 = (tile % tiles_n) * 256 + wc * 64;
        u32x4 v = {(unsigned)tile, (unsigned)lane, 3u, 4u};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int m, n;
            if (PAT == 0) { m = m0 + 16 * (i >> 1) + r16; n = n0 + 32 * (i & 1) + 8 * g; }
            else if (PAT == 1 || PAT == 3) { m = m0 + 8 * i + (lane >> 3); n = n0 + 8 * (lane & 7); }
            else { m = m0 + 8 * i + (lane >> 3); n = n0 + 8 * (lane & 7); }   // (pattern 2 below)
            unsigned short* p = C + (size_t)m * ldc + n;
            if (PAT == 2) {          // 4 rows x 256 bytes: two neighbouring wave columns' ranges interleaved
                m = m0 + 4 * i + (lane >> 4) + ((wc & 1) ? 64 : 0);
                n = (tile % tiles_n) * 256 + (wc >> 1) * 128 + 8 * (lane & 15);
                p = C + (size_t)m * ldc + n;
            }
            if (PAT == 3) __builtin_nontemporal_store(v, (u32x4*)p);
            else *(u32x4*)p = v;
            v.x += 1;
        }
    }
}

int main() {
    const int M = 65536, N = 1536, tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
    unsigned short* C;
    hipMalloc(&C, (size_t)M * N * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // per-CU limit or fabric limit?  pattern 0 on fewer workgroups, and on a small (L2-resident) footprint
    for (int grid : {256, 128, 64, 32, 8}) {
        for (int small = 0; small < 2; ++small) {
            float ms = 0;
            const int tpw = 24;
            const int nt = small ? 32 : ntiles;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(wr<0>, dim3(grid), dim3(512), 0, 0, C, N, tiles_n, nt, tpw);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double bytes = (double)grid * tpw * 131072.0;
            printf("grid %3d %s: %.1f us  %.2f TB/s  %.1f GB/s per CU = %.1f B/clk@2.4GHz\n", grid, small ? "4MB footprint" : "full footprint",
                   ms * 1000, bytes / ms / 1e9, bytes / ms / 1e6 / grid, bytes / ms / 1e6 / grid / 2.4);
        }
    }
    return 0;
}
