// De-risking a fused key-stationary attention backward: what do its two new memory patterns cost at the bench
// shape?  (a) dq accumulated with fp32 atomics: per (batch, head) pair, workgroups for 4 key tiles of 256 keys add [32 queries x 64
// features] partials (8 KB, 256 contiguous bytes per wave instruction) into the pair's rows of a [T*B, 512] fp32 buffer --
// 80 partials per pair, 512 pairs, 0.33 GB per layer.  (b) dS by distance written straight from a lane = key register layout: 16
// two-byte stores per 32 x 32 block (each: two runs of 32 consecutive distances = 64 contiguous bytes), 0.57 GB per layer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
typedef __attribute__((ext_vector_type(4))) float f4;
constexpr int T = 1024, B = 64, H = 8, HD = 512;
__global__ __launch_bounds__(256) void atomic_dq(float* dq, int rounds) {
    // workgroup = (pair, key tile kt of 256 keys): query blocks of 32 rows from 256 kt to T
    const int pair = blockIdx.x >> 2, kt = blockIdx.x & 3, b = pair / H, h = pair % H;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = 0; r < rounds; ++r)
        for (int i0 = 256 * kt; i0 < T; i0 += 32) {
            // 4 waves x 8 rows x 64 features: lane = feature, one row per instruction
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int i = i0 + 8 * w + rr;
                __hip_atomic_fetch_add(dq + ((size_t)i * B + b) * HD + h * 64 + lane, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
}
__global__ __launch_bounds__(256) void store_dq(float* dq, int rounds) {          // the same pattern with plain stores (reference)
    const int pair = blockIdx.x >> 2, kt = blockIdx.x & 3, b = pair / H, h = pair % H;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = 0; r < rounds; ++r)
        for (int i0 = 256 * kt; i0 < T; i0 += 32)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int i = i0 + 8 * w + rr;
                dq[((size_t)i * B + b) * HD + h * 64 + lane] = 1.0f + r;
            }
}
__global__ __launch_bounds__(256) void short_stores(unsigned short* dsk, int ld) {
    // workgroup = (pair, key tile of 128 keys: 4 waves x 32 keys); per 32-query block 16 stores of one bf16 per lane:
    // register r = query 8 (r >> 2) + 4 half + (r & 3) of the block, lane = key -> distance i + M - j (tiled [64 rows][128 d] layout)
    const int pair = blockIdx.x >> 3, kt = blockIdx.x & 7, b = pair / H, h = pair % H;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, key = lane & 31, half = lane >> 5;
    const int j = 128 * kt + 32 * w + key;
    unsigned short* base = dsk + (size_t)h * T * B * ld;
    for (int i0 = 128 * kt; i0 < T; i0 += 32)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + 8 * (r >> 2) + 4 * half + (r & 3), d = i - j;
            if (d >= 0) {
                const size_t m = (size_t)i * B + b;
                base[(((m >> 6) * (ld >> 7) + (d >> 7)) << 13) + ((m & 63) << 7) + (d & 127)] = (unsigned short)r;
            }
        }
}
int main() {
    float* dq; unsigned short* dsk;
    hipMalloc(&dq, (size_t)T * B * HD * 4);
    hipMalloc(&dsk, (size_t)H * T * B * 1024 * 2);
    hipMemset(dq, 0, (size_t)T * B * HD * 4);
    auto run = [&](const char* name, auto fn) {
        fn(); hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 5; ++k) fn();
        hipDeviceSynchronize();
        printf("%-28s %8.1f us per launch\n", name, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 5);
    };
    run("atomic dq (0.33 GB)", [&] { atomic_dq<<<B * H * 4, 256>>>(dq, 1); });
    run("plain stores, same pattern", [&] { store_dq<<<B * H * 4, 256>>>(dq, 1); });
    run("2-byte dS stores (0.57 GB)", [&] { short_stores<<<B * H * 8, 256>>>(dsk, 1024); });
    return 0;
}
