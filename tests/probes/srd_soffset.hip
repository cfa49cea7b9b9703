// Does gfx950 range-check the SCALAR offset of a raw buffer load?  gemm_nt8 (csrc/gemm8.hip) builds one descriptor per
// operand and puts the tile's first row into soffset; the zero-fill of rows beyond M / N in edge tiles then depends on the
// hardware comparing voffset + soffset (not voffset alone) with num_records.  LLVM documents soffset as excluded from the
// bounds check of raw buffers; this probe settles it on the part: a 4-KB buffer of ones, a descriptor that covers its first
// 1 KB, loads at (voffset, soffset) pairs on both sides of the limit.  Prints one line per case and a verdict.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/srd_soffset tests/probes/srd_soffset.hip && /tmp/srd_soffset
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __amdgpu_buffer_rsrc_t srd_t;

__global__ void probe(const unsigned* buf, unsigned* out, unsigned* out_lds) {
    __shared__ unsigned sm[64 * 4];
    const srd_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 1024, 0x00020000);
    // case 0: voffset 512, soffset 0 (inside); 1: voffset 2048, soffset 0 (outside via voffset);
    // case 2: voffset 0, soffset 2048 (outside via soffset only); 3: voffset 512, soffset 768 (sum outside, each inside)
    const unsigned vo[4] = {512u, 2048u, 0u, 512u}, so[4] = {0u, 0u, 2048u, 768u};
    for (int c = 0; c < 4; ++c) {
        out[c] = __builtin_amdgcn_raw_buffer_load_b32(srd, vo[c], so[c], 0);
        // the same through LDS-DMA (buffer_load_dword ... lds), the path gemm_nt8 uses
        sm[threadIdx.x] = 0xDEADu;
        __syncthreads();
        const unsigned lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)sm;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds\n\ts_waitcnt vmcnt(0)"
                     :: "s"(lds_dst), "v"(vo[c] + 4u * threadIdx.x * 0u), "s"(srd), "s"(so[c]) : "memory");
        __syncthreads();
        out_lds[c] = sm[0];
        __syncthreads();
    }
}

int main() {
    unsigned *buf, *out, *out_lds, h[4096 / 4], r[4], rl[4];
    for (auto& x : h) x = 1u;
    hipMalloc(&buf, 4096); hipMalloc(&out, 16); hipMalloc(&out_lds, 16);
    hipMemcpy(buf, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, out, out_lds);
    hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
    hipMemcpy(rl, out_lds, 16, hipMemcpyDeviceToHost);
    const char* name[4] = {"inside (v 512, s 0)", "outside by voffset (v 2048, s 0)", "outside by soffset (v 0, s 2048)",
                           "outside by the sum (v 512, s 768)"};
    for (int c = 0; c < 4; ++c) printf("%-36s register load -> %u   lds-dma -> %u\n", name[c], r[c], rl[c]);
    const bool checked = r[0] == 1 && r[1] == 0 && r[2] == 0 && r[3] == 0 && rl[0] == 1 && rl[1] == 0 && rl[2] == 0 && rl[3] == 0;
    printf("verdict: soffset %s in the range check of raw buffer loads on this part\n", checked ? "IS INCLUDED" : "is NOT included");
    return checked ? 0 : 3;
}
