// Probe: empirical lane/element mapping of ds_read_b64_tr_b16 on gfx950.
// Each lane supplies address base + lane*8 bytes; LDS holds element index as value.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out, int variant) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    int l = threadIdx.x;
    int idx = (variant == 0) ? l * 4                                   // lane-linear
                             : ((l >> 4) * 8 + ((l & 15) >> 2)) * 128 + 4 * (l & 3);   // [row = 8g + i/4][col 4*(i&3)], pitch 128
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + idx));
    for (int j = 0; j < 4; j++) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    for (int variant = 0; variant < 2; ++variant) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, variant);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("variant %d\n", variant);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
