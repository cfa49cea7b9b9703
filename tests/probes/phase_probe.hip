// Micro-benchmark of the phases of one 16x64 attention wave tile, in isolation and combined, at 1..4 waves per SIMD:
// where do the ~3100 cycles per wave tile per SIMD of the forward kernel go?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define LDS_AS __attribute__((address_space(3)))
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
__device__ __forceinline__ float bperm(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}
// MODE bits: 1 MFMA (26), 2 skew (16 cndmask + 16 bpermute), 4 softmax VALU (max, exp, sum, cvt), 8 LDS frag reads (18 b128 + 20 tr)
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, int pad) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];      // 40 KB tile area (+ dynamic padding to set occupancy)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    for (int i = threadIdx.x; i < 20480; i += 256) smem[i] = (__bf16)(0.001f * (i & 127));
    __syncthreads();
    bf16x8 qa, qb;
    for (int e = 0; e < 8; ++e) { qa[e] = (__bf16)(0.01f * (lane + e)); qb[e] = (__bf16)(0.02f * (lane - e)); }
    f32x4 o[4], s[4], qr[5];
    for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < 4; ++c) s[c] = (f32x4){0.1f * lane, 0.2f, 0.3f, 0.4f};
    for (int c = 0; c < 5; ++c) qr[c] = (f32x4){0.01f * lane, 0.02f, 0.03f, 0.04f};
    float mrow[4] = {-1e30f, -1e30f, -1e30f, -1e30f}, lpart[4] = {0.f, 0.f, 0.f, 0.f};
    int srcaddr[4];
    bool lower[4];
    for (int reg = 0; reg < 4; ++reg) { srcaddr[reg] = ((lane & 48) | ((4 * g + reg - 1 - r16) & 15)) << 2; lower[reg] = r16 < 4 * g + reg; }
    bf16x8 kf[8], rf[10];
    for (int i = 0; i < 8; ++i) kf[i] = qa;
    for (int i = 0; i < 10; ++i) rf[i] = qb;
    const __bf16* sK = smem, *sR = smem + 4096, *sV = smem + 12288;
    __bf16* myP = smem + 16384 + w * 1024;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 8) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    kf[2 * c + ks] = *(const bf16x8*)(sK + (16 * c + r16) * 64 + (((4 * ks + g) ^ (r16 & 7)) << 3));
#pragma unroll
            for (int b = 0; b < 5; ++b)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    rf[2 * b + ks] = *(const bf16x8*)(sR + (((16 * w + 16 * b + it * 64) & 127) + r16) * 64 + (((4 * ks + g) ^ (r16 & 7)) << 3));
        }
        if (MODE & 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 5; ++b) qr[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int c = 0; c < 4; ++c) s[c] = mfma16(qa, kf[2 * c + ks], s[c]);
#pragma unroll
                for (int b = 0; b < 5; ++b) qr[b] = mfma16(qb, rf[2 * b + ks], qr[b]);
            }
        }
        if (MODE & 2) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float t0 = lower[reg] ? qr[4][reg] : qr[3][reg], t1 = lower[reg] ? qr[3][reg] : qr[2][reg],
                            t2 = lower[reg] ? qr[2][reg] : qr[1][reg], t3 = lower[reg] ? qr[1][reg] : qr[0][reg];
                s[0][reg] += bperm(srcaddr[reg], t0);
                s[1][reg] += bperm(srcaddr[reg], t1);
                s[2][reg] += bperm(srcaddr[reg], t2);
                s[3][reg] += bperm(srcaddr[reg], t3);
            }
        }
        if (MODE & 4) {
            float mnew[4];
            bool grew = false;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float mx = fmaxf(fmaxf(s[0][reg], s[1][reg]), fmaxf(s[2][reg], s[3][reg]));
                mx = row16_max(mx);
                mnew[reg] = fmaxf(mrow[reg], mx);
                grew |= mnew[reg] > mrow[reg];
            }
            if (__any(grew)) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float alpha = __builtin_amdgcn_exp2f(mrow[reg] - mnew[reg]);
                    mrow[reg] = mnew[reg];
                    lpart[reg] *= alpha;
#pragma unroll
                    for (int d = 0; d < 4; ++d) o[d][reg] *= alpha;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bf16x4 pb;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float p = __builtin_amdgcn_exp2f(s[c][reg] - mrow[reg]);
                    lpart[reg] += p;
                    pb[reg] = (__bf16)p;
                }
                *(bf16x4*)(myP + (16 * c + r16) * 16 + 4 * g) = pb;
            }
        }
        if (MODE & 8) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 pf;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bf16x4 v = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (LDS_AS s16x4*)(myP + (32 * ks + 8 * g + 4 * h + (r16 >> 2)) * 16 + 4 * (r16 & 3))));
                    for (int e = 0; e < 4; ++e) pf[4 * h + e] = v[e];
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    bf16x8 vf;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = 32 * ks + 8 * g + 4 * h + (r16 >> 2), col = 16 * d + 4 * (r16 & 3);
                        const bf16x4 v = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (LDS_AS s16x4*)(sV + row * 64 + (((col >> 3) ^ (row & 7)) << 3) + (col & 7))));
                        for (int e = 0; e < 4; ++e) vf[4 * h + e] = v[e];
                    }
                    if (MODE & 1) o[d] = mfma16(pf, vf, o[d]);
                    else o[d][0] += (float)vf[0] + (float)pf[1];
                }
            }
        } else if (MODE & 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int d = 0; d < 4; ++d) o[d] = mfma16(qa, qb, o[d]);
        }
    }
    float acc = 0.f;
    for (int d = 0; d < 4; ++d) for (int e = 0; e < 4; ++e) acc += o[d][e] + s[d][e];
    for (int e = 0; e < 4; ++e) acc += lpart[e] + mrow[e] + qr[4][e];
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

// "all" with the fragment reads of tile t+1 (K and band) issued right after tile t's QK/QR MFMAs, and tile t's V fragments
// requested before the softmax: every LDS round trip sits under arithmetic.  PRE: 0 = none, 1 = V early, 2 = V early + next K/R
template <int PRE, bool LAZY = false>
__global__ __launch_bounds__(256) void probe_pipe(float* out, int iters, int pad) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    for (int i = threadIdx.x; i < 20480; i += 256) smem[i] = (__bf16)(0.001f * (i & 127));
    __syncthreads();
    bf16x8 qa, qb;
    for (int e = 0; e < 8; ++e) { qa[e] = (__bf16)(0.01f * (lane + e)); qb[e] = (__bf16)(0.02f * (lane - e)); }
    f32x4 o[4], s[4], qr[5];
    for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrow[4] = {-1e30f, -1e30f, -1e30f, -1e30f}, lpart[4] = {0.f, 0.f, 0.f, 0.f};
    int srcaddr[4];
    bool lower[4];
    for (int reg = 0; reg < 4; ++reg) { srcaddr[reg] = ((lane & 48) | ((4 * g + reg - 1 - r16) & 15)) << 2; lower[reg] = r16 < 4 * g + reg; }
    const __bf16* sK = smem, *sR = smem + 4096, *sV = smem + 12288;
    __bf16* myP = smem + 16384 + w * 1024;
    bf16x8 kf[8], rf[10];
    auto load_kr = [&](int it) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                kf[2 * c + ks] = *(const bf16x8*)(sK + (16 * c + r16) * 64 + (((4 * ks + g) ^ (r16 & 7)) << 3));
#pragma unroll
        for (int b = 0; b < 5; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                rf[2 * b + ks] = *(const bf16x8*)(sR + (((16 * w + 16 * b + it * 64) & 127) + r16) * 64 + (((4 * ks + g) ^ (r16 & 7)) << 3));
    };
    load_kr(0);
    for (int it = 0; it < iters; ++it) {
        if (PRE < 2) load_kr(it);
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 5; ++b) qr[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] = mfma16(qa, kf[2 * c + ks], s[c]);
#pragma unroll
            for (int b = 0; b < 5; ++b) qr[b] = mfma16(qb, rf[2 * b + ks], qr[b]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PRE >= 2) load_kr(it + 1);                       // next tile's operands fly under skew + softmax + P.V
        bf16x8 vf[2][4];
        if (PRE >= 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = 32 * ks + 8 * g + 4 * h + (r16 >> 2), col = 16 * d + 4 * (r16 & 3);
                        const bf16x4 v = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (LDS_AS s16x4*)(sV + row * 64 + (((col >> 3) ^ (row & 7)) << 3) + (col & 7))));
                        for (int e = 0; e < 4; ++e) vf[ks][d][4 * h + e] = v[e];
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float t0 = lower[reg] ? qr[4][reg] : qr[3][reg], t1 = lower[reg] ? qr[3][reg] : qr[2][reg],
                        t2 = lower[reg] ? qr[2][reg] : qr[1][reg], t3 = lower[reg] ? qr[1][reg] : qr[0][reg];
            s[0][reg] += bperm(srcaddr[reg], t0);
            s[1][reg] += bperm(srcaddr[reg], t1);
            s[2][reg] += bperm(srcaddr[reg], t2);
            s[3][reg] += bperm(srcaddr[reg], t3);
        }
        if (LAZY) {
            // scores already carry "- m" (folded into the accumulator initialisation): only test for growth
            bool big = false;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                big |= fmaxf(fmaxf(s[0][reg], s[1][reg]), fmaxf(s[2][reg], s[3][reg])) > 6.f;
            if (__any(big)) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    float mx = fmaxf(fmaxf(s[0][reg], s[1][reg]), fmaxf(s[2][reg], s[3][reg]));
                    mx = fmaxf(row16_max(mx), 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-mx);
                    mrow[reg] += mx;
                    lpart[reg] *= alpha;
#pragma unroll
                    for (int d = 0; d < 4; ++d) o[d][reg] *= alpha;
#pragma unroll
                    for (int c = 0; c < 4; ++c) s[c][reg] -= mx;
                }
            }
        } else {
        float mnew[4];
        bool grew = false;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float mx = fmaxf(fmaxf(s[0][reg], s[1][reg]), fmaxf(s[2][reg], s[3][reg]));
            mx = row16_max(mx);
            mnew[reg] = fmaxf(mrow[reg], mx);
            grew |= mnew[reg] > mrow[reg];
        }
        if (__any(grew)) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float alpha = __builtin_amdgcn_exp2f(mrow[reg] - mnew[reg]);
                mrow[reg] = mnew[reg];
                lpart[reg] *= alpha;
#pragma unroll
                for (int d = 0; d < 4; ++d) o[d][reg] *= alpha;
            }
        }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x4 pb;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float p = LAZY ? __builtin_amdgcn_exp2f(s[c][reg]) : __builtin_amdgcn_exp2f(s[c][reg] - mrow[reg]);
                lpart[reg] += p;
                pb[reg] = (__bf16)p;
            }
            *(bf16x4*)(myP + (16 * c + r16) * 16 + 4 * g) = pb;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 pf;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x4 v = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (LDS_AS s16x4*)(myP + (32 * ks + 8 * g + 4 * h + (r16 >> 2)) * 16 + 4 * (r16 & 3))));
                for (int e = 0; e < 4; ++e) pf[4 * h + e] = v[e];
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x8 v8;
                if (PRE >= 1) v8 = vf[ks][d];
                else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = 32 * ks + 8 * g + 4 * h + (r16 >> 2), col = 16 * d + 4 * (r16 & 3);
                        const bf16x4 v = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (LDS_AS s16x4*)(sV + row * 64 + (((col >> 3) ^ (row & 7)) << 3) + (col & 7))));
                        for (int e = 0; e < 4; ++e) v8[4 * h + e] = v[e];
                    }
                }
                o[d] = mfma16(pf, v8, o[d]);
            }
        }
    }
    float acc = 0.f;
    for (int d = 0; d < 4; ++d) for (int e = 0; e < 4; ++e) acc += o[d][e] + s[d][e];
    for (int e = 0; e < 4; ++e) acc += lpart[e] + mrow[e] + qr[4][e] + (float)kf[0][e] + (float)rf[9][e];
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}
template <int PRE, bool LAZY = false> void run_pipe(const char* name, float* d_out) {
    const int iters = 2000;
    for (int occ = 1; occ <= 3; ++occ) {
        const int lds = occ == 1 ? 150 * 1024 : (occ == 2 ? 78 * 1024 : 52 * 1024);
        hipFuncSetAttribute((const void*)probe_pipe<PRE, LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        probe_pipe<PRE, LAZY><<<256 * occ, 256, lds>>>(d_out, 10, 0);
        hipDeviceSynchronize();
        hipEventRecord(a);
        probe_pipe<PRE, LAZY><<<256 * occ, 256, lds>>>(d_out, iters, 0);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-28s waves/SIMD %d: %8.1f cycles per tile per wave, %8.1f per tile per SIMD\n", name, occ,
               ms * 1e-3 * 2.4e9 / iters, ms * 1e-3 * 2.4e9 / iters / occ);
    }
}

template <int MODE> void run(const char* name, float* d_out) {
    const int iters = 2000;
    for (int occ = 1; occ <= 4; ++occ) {
        // dynamic LDS sets the workgroups per CU: 160 KB / occ (minus a little), each workgroup = 4 waves = 1 wave per SIMD
        const int lds = occ == 1 ? 150 * 1024 : (occ == 2 ? 78 * 1024 : (occ == 3 ? 52 * 1024 : 40 * 1024));
        hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const int grid = 256 * occ;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        probe<MODE><<<grid, 256, lds>>>(d_out, 10, 0);
        hipDeviceSynchronize();
        hipEventRecord(a);
        probe<MODE><<<grid, 256, lds>>>(d_out, iters, 0);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // cycles per wave tile per SIMD at 2.4 GHz: each SIMD runs `occ` waves, each doing `iters` tiles
        printf("%-28s waves/SIMD %d: %8.1f cycles per tile per wave, %8.1f per tile per SIMD\n", name, occ,
               ms * 1e-3 * 2.4e9 / iters, ms * 1e-3 * 2.4e9 / iters / occ);
    }
}
int main() {
    float* d_out; hipMalloc(&d_out, 4096);
    run<1>("MFMA only (26)", d_out);
    run<2>("skew only", d_out);
    run<4>("softmax VALU + P write", d_out);
    run<8>("LDS fragment reads only", d_out);
    run<3>("MFMA + skew", d_out);
    run<7>("MFMA + skew + softmax", d_out);
    run<10>("skew + LDS reads", d_out);
    run<14>("skew + softmax + LDS reads", d_out);
    run<9>("MFMA + LDS reads", d_out);
    run<12>("softmax + LDS reads", d_out);
    run<11>("MFMA + skew + LDS reads", d_out);
    run<13>("MFMA + softmax + LDS reads", d_out);
    run<15>("all (no global, no barrier)", d_out);
    run_pipe<0>("pipe: as shipped", d_out);
    run_pipe<1>("pipe: V early", d_out);
    run_pipe<2>("pipe: V early + next K/R", d_out);
    run_pipe<0, true>("lazy max, folded subtraction", d_out);
    return 0;
}
