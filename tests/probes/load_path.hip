// Probe: bytes per clock a CU pulls from L2 (a) by LDS-DMA (buffer_load_dwordx4 ... lds), (b) by global_load_dwordx4 into
// registers, (c) as (b) followed by ds_write_b128 -- the three ways a GEMM K-tile can reach LDS.  256 workgroups x 512
// threads; every workgroup re-reads its own REGION bytes (L2-resident) ITERS times, DEPTH wave-loads in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 load_path.hip -o bin/load_path && bin/load_path
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __amdgpu_buffer_rsrc_t srd_t;
#define LDS_AS __attribute__((address_space(3)))

constexpr int REGION = 64 * 1024;          // bytes per workgroup and pass (one K-tile of a 256 x 256 x 64 GEMM step)

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k_load(const char* __restrict__ src, unsigned* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * REGION];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * REGION;
    const srd_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, REGION, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    // one pass = REGION / (512 * 16) = 8 loads per thread
    for (int it = 0; it < iters; ++it) {
        const unsigned pb = (it & 1) * REGION;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned dst = lds0 + pb + j * 8192 + w * 1024;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                             :: "s"(dst), "v"((unsigned)((j * 8192 + tid * 16 + ((unsigned)(it * 4096) & (REGION - 1))) & (REGION - 1))), "s"(srd) : "memory");
                if ((j % DEPTH) == DEPTH - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 v[8];
            const unsigned rot = (unsigned)(it * 4096) & (REGION - 1);          // (a different address every pass: nothing to hoist)
#pragma unroll
            for (int j0 = 0; j0 < 8; j0 += DEPTH) {
#pragma unroll
                for (int j = j0; j < j0 + DEPTH && j < 8; ++j)
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen"
                                 : "=v"(v[j]) : "v"((unsigned)((j * 8192 + tid * 16 + rot) & (REGION - 1))), "s"(srd) : "memory");
#pragma unroll
                for (int j = j0; j < j0 + DEPTH && j < 8; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[j]) :: "memory");
#pragma unroll
                for (int j = j0; j < j0 + DEPTH && j < 8; ++j) {
                    if (MODE == 2) *(u32x4*)(smem + pb + j * 8192 + tid * 16) = v[j];
                    else acc ^= v[j];
                }
            }
            asm volatile("" :: "v"(acc) : "memory");
        }
        __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (MODE != 1) acc[0] ^= *(unsigned*)(smem + tid * 4);
    if (acc[0] == 0x12345u) out[blockIdx.x * 512 + tid] = acc[1];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int DEPTH>
void run(const char* name, const char* src, unsigned* out, long long* cyc) {
    const int iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_load<MODE, DEPTH>), dim3(256), dim3(512), 0, 0, src, out, 20, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k_load<MODE, DEPTH>), dim3(256), dim3(512), 0, 0, src, out, iters, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)h[i]; mean /= 256;
    const double bytes = (double)REGION * iters;
    printf("%-34s depth %d: %7.1f us, %6.2f TB/s chip, %5.1f B per shader clock per CU (cycle counter: %.0f cycles per pass)\n", name, DEPTH,
           ms * 1e3, bytes * 256 / (ms * 1e-3) / 1e12, bytes / mean / 1.0, mean / iters);
}

int main() {
    char* src; unsigned* out; long long* cyc;
    hipMalloc(&src, (size_t)256 * REGION); hipMemset(src, 1, (size_t)256 * REGION);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    run<0, 2>("LDS-DMA (buffer_load ... lds)", src, out, cyc);
    run<0, 4>("LDS-DMA (buffer_load ... lds)", src, out, cyc);
    run<0, 8>("LDS-DMA (buffer_load ... lds)", src, out, cyc);
    run<1, 2>("global_load_dwordx4 -> VGPR", src, out, cyc);
    run<1, 4>("global_load_dwordx4 -> VGPR", src, out, cyc);
    run<1, 8>("global_load_dwordx4 -> VGPR", src, out, cyc);
    run<2, 4>("global_load -> VGPR -> ds_write", src, out, cyc);
    run<2, 8>("global_load -> VGPR -> ds_write", src, out, cyc);
    return 0;
}
