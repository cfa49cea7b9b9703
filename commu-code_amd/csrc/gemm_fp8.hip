// MX-fp8 NT GEMM for gfx950 (BASELINE.json configs[4]: "fp8 MFMA GEMMs"):  C[M,N] = A[M,K] . B[N,K]^T  with A and B
// in OCP e4m3 and one E8M0 scale per 32 consecutive k (the OCP microscaling format the CDNA4 matrix core consumes
// directly: v_mfma_scale_f32_16x16x128_f8f6f4 -- 128 k per instruction, twice the bf16 rate), fp32 accumulation,
// bf16 output.  Same row-major "NT" convention as commu_gemm_nt_bf16 (activations [tokens, K], nn.Linear weight [N, K]).
//
//  * commu_quant_mxfp8: bf16 [rows, K] -> e4m3 bytes [rows, K] + scale bytes [rows, K/32].  Shared exponent of a block
//    = floor(log2(amax)) - 8 (e4m3's largest power of two), elements round to nearest even and saturate at +-448: the
//    OCP MX v1.0 recipe.  HBM-bound: 2 B read + 1.03 B written per element.
//  * commu_gemm_nt_mxfp8: 128 x 128 output tile per workgroup (4 waves, 2 x 2, 64 x 64 each = 4 x 4 MFMA tiles), K in
//    steps of 128 bytes, both operand tiles staged through LDS (XOR-swizzled 16-byte chunks, double-buffered, buffer
//    loads with out-of-range rows reading zero), scales read per (row, k-step) as one dword.  The epilogue goes through
//    LDS so that C leaves as 16-byte chunks of whole rows.
//
// Lane layout of the scaled MFMA (probed on the device, tests/probes/fp8probe.py, and checked against the CPU
// emulation in tests/test_fp8_gemm_gpu.py with asymmetric operands): lane l holds row l & 15 of its operand; with
// g = l >> 4 its first four dwords are k = 16 g .. 16 g + 15 and its last four k = 64 + 16 g .. 64 + 16 g + 15, while the
// scale operand of lane group g (byte 0, op_sel 0) is the row's scale of the 32-block k = 32 g .. 32 g + 31.
#include "common.h"
#include "commu_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __amdgpu_buffer_rsrc_t srd8_t;

__device__ __forceinline__ srd8_t mk_srd8(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(unsigned)bytes, 0x00020000);
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_mxfp8_kernel(const bf16* __restrict__ X, int ldx, unsigned char* __restrict__ Q,
                                                          int ldq, unsigned char* __restrict__ S, int lds, int rows, int K) {
    const int cpr = K >> 3;                                        // 8-element chunks per row
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < (long long)rows * cpr;
    const int r = live ? (int)(idx / cpr) : 0, c = live ? (int)(idx % cpr) : 0;
    float v[8];
    float amax = 0.f;
    if (live) {
        const bf16x8 raw = ld_bf16x8(X + (size_t)r * ldx + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = bf2f(raw[e]);
            amax = fmaxf(amax, fabsf(v[e]));
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    // a 32-block is four consecutive lanes (K % 32 == 0, so blocks never straddle rows or quads)
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    const int eb = (int)((__float_as_uint(amax) >> 23) & 0xFFu);  // biased floor(log2 amax) (0 for zero / denormal blocks)
    int sb = eb - 8;                                               // E8M0 byte: scale 2^(sb - 127)
    sb = sb < 0 ? 0 : (sb > 254 ? 254 : sb);
    const float inv = __uint_as_float((unsigned)(254 - sb) << 23); // 2^(127 - sb)
    int w0 = 0, w1 = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(v[e] * inv, -448.f), 448.f);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
    if (live) {
        *(i32x2*)(Q + (size_t)r * ldq + 8 * c) = (i32x2){w0, w1};
        if ((c & 3) == 0) S[(size_t)r * lds + (c >> 2)] = (unsigned char)sb;
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct F8Args {
    const unsigned char *A, *SA, *B, *SB;
    bf16* C;
    const float* bias;
    const bf16* resid;
    int lda, ldsa, ldb, ldsb, ldc, ldr, M, N, K, flags;
    unsigned drop_seed, drop_thr;
    float drop_scale;
};

constexpr int TM = 128, TN = 128, TK = 128;          // TK in bytes = k elements
constexpr int TILE_B = TM * TK;                      // 16 KB per operand tile

__global__ __launch_bounds__(256) void gemm_nt_mxfp8_kernel(const F8Args a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TILE_B];          // [2 buffers][A | B], 64 KB
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    // XCD-aware tile order: consecutive tiles of one XCD share the B (weight) panel in its L2
    const int tiles_m = (a.M + TM - 1) / TM, tiles_n = (a.N + TN - 1) / TN;
    const int tile = xcd_remap((int)blockIdx.x, tiles_m * tiles_n);
    const int tn = tile / tiles_m, tm = tile - tn * tiles_m;
    const int m0 = tm * TM, n0 = tn * TN;

    const srd8_t srdA = mk_srd8(a.A, (size_t)(a.M - 1) * a.lda + a.K);
    const srd8_t srdB = mk_srd8(a.B, (size_t)(a.N - 1) * a.ldb + a.K);
    // staging: 1024 16-byte chunks per operand tile, four per thread: chunk q = tid + 256 n -> row q >> 3, chunk q & 7
    unsigned goA[4], goB[4], lo[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int q = tid + 256 * n, row = q >> 3, ch = q & 7;
        goA[n] = (unsigned)(m0 + row) * (unsigned)a.lda + 16u * ch;          // rows past M / N: out of range -> zeros
        goB[n] = (unsigned)(n0 + row) * (unsigned)a.ldb + 16u * ch;
        if (m0 + row >= a.M) goA[n] = 0x80000000u;          // (stays out of range for every k offset added later)
        if (n0 + row >= a.N) goB[n] = 0x80000000u;
        lo[n] = (unsigned)(row * TK + ((ch ^ (row & 7)) << 4));
    }
    i32x4 ra[4], rb[4];
    auto gload = [&](int kk) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            ra[n] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(srdA, (int)(goA[n] + (unsigned)kk), 0, 0));
            rb[n] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(srdB, (int)(goB[n] + (unsigned)kk), 0, 0));
        }
    };
    auto commit = [&](int buf) {
        unsigned char* dA = smem + buf * 2 * TILE_B;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            *(i32x4*)(dA + lo[n]) = ra[n];
            *(i32x4*)(dA + TILE_B + lo[n]) = rb[n];
        }
    };
    // fragment addressing: row 64 wm + 16 i + r16 of the tile, 16-byte chunks g (k = 16g ..) and 4 + g (k = 64 + 16g ..)
    int fo[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) fo[hf] = r16 * TK + (((g + 4 * hf) ^ (r16 & 7)) << 4);
    // scales: one dword per (row, k-step); this lane needs byte g of it
    const unsigned char* sa[4];
    const unsigned char* sb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sa[i] = a.SA + (size_t)min(m0 + 64 * wm + 16 * i + r16, a.M - 1) * a.ldsa;
        sb[i] = a.SB + (size_t)min(n0 + 64 * wn + 16 * i + r16, a.N - 1) * a.ldsb;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / TK;
    gload(0);
    commit(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * TK);
        const unsigned char* tA = smem + buf * 2 * TILE_B + (64 * wm) * TK;
        const unsigned char* tB = smem + buf * 2 * TILE_B + TILE_B + (64 * wn) * TK;
        int sca[4], scb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sca[i] = (int)((*(const unsigned*)(sa[i] + 4 * kt) >> (8 * g)) & 0xFFu);
            scb[i] = (int)((*(const unsigned*)(sb[i] + 4 * kt) >> (8 * g)) & 0xFFu);
        }
        i32x8 fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i32x4 x0 = *(const i32x4*)(tB + 16 * j * TK + fo[0]), x1 = *(const i32x4*)(tB + 16 * j * TK + fo[1]);
            fb[j] = (i32x8){x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const i32x4 x0 = *(const i32x4*)(tA + 16 * i * TK + fo[0]), x1 = *(const i32x4*)(tA + 16 * i * TK + fo[1]);
            const i32x8 fa = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb[j], acc[i][j], 0, 0, 0, sca[i], 0, scb[j]);
        }
        if (kt + 1 < nk) commit(buf ^ 1);          // the other buffer was last read before the previous barrier
        __syncthreads();
    }
    // epilogue: the wave's 64 x 64 block through its own 8 KB of LDS (bf16, row pitch 64 + 8 elements), out as 16-byte
    // chunks of whole rows; bias and ReLU applied on the way
    bf16* ep = (bf16*)smem + w * (64 * 72);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + 64 * wn + 16 * j + r16;
            const float bv = ((a.flags & COMMU_EPI_BIAS) && col < a.N) ? a.bias[col] : 0.f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                // same order and the same dropout element index (m * N + n) as commu_gemm_nt_bf16: the backward pass
                // regenerates the masks from (seed, index)
                const int row = m0 + 64 * wm + 16 * i + 4 * g + reg;
                float x = acc[i][j][reg] + bv;
                if (a.flags & COMMU_EPI_RELU) x = fmaxf(x, 0.f);
                if (a.flags & COMMU_EPI_DROPOUT)
                    x = drop_keep(salted(a.drop_seed), (unsigned)row * (unsigned)a.N + (unsigned)col, a.drop_thr) ? x * a.drop_scale : 0.f;
                if ((a.flags & COMMU_EPI_RESID) && row < a.M && col < a.N) x += bf2f(a.resid[(size_t)row * a.ldr + col]);
                ep[(16 * i + 4 * g + reg) * 72 + 16 * j + r16] = f2bf(x);
            }
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int row = (lane >> 3) + 8 * n, ch = lane & 7;
        const int gr = m0 + 64 * wm + row, gc = n0 + 64 * wn + 8 * ch;
        if (gr < a.M && gc + 8 <= a.N) st_bf16x8(a.C + (size_t)gr * a.ldc + gc, ld_bf16x8(ep + row * 72 + 8 * ch));
        else if (gr < a.M)
            for (int e = 0; e < 8 && gc + e < a.N; ++e) a.C[(size_t)gr * a.ldc + gc + e] = ep[row * 72 + 8 * ch + e];
    }
}

}  // namespace

extern "C" int commu_quant_mxfp8(const void* X, int ldx, void* Q, int ldq, void* S, int lds, int rows, int K,
                                 hipStream_t stream) {
    if (rows <= 0 || K <= 0) return 0;
    if ((K % 32) || (ldx % 8) || (ldq % 8) || ldq < K || lds < K / 32) return -22;
    const long long chunks = (long long)rows * (K / 8);
    COMMU_LAUNCH(quant_mxfp8_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream, (const bf16*)X, ldx,
                 (unsigned char*)Q, ldq, (unsigned char*)S, lds, rows, K);
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_gemm_nt_mxfp8(const void* A, int lda, const void* SA, int ldsa, const void* B, int ldb, const void* SB,
                                   int ldsb, void* C, int ldc, int M, int N, int K, const float* bias, const void* resid,
                                   int ldr, int flags, unsigned drop_seed, float drop_p, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K % 128) || (lda % 16) || (ldb % 16) || (ldsa % 4) || (ldsb % 4) || (ldc % 8) || lda < K || ldb < K ||
        ldsa < K / 32 || ldsb < K / 32)
        return -22;
    if ((size_t)M * lda >= 0x7FFF0000ull || (size_t)N * ldb >= 0x7FFF0000ull) return -22;
    if (flags & ~(COMMU_EPI_BIAS | COMMU_EPI_RELU | COMMU_EPI_RESID | COMMU_EPI_DROPOUT)) return -22;
    if (((flags & COMMU_EPI_BIAS) && !bias) || ((flags & COMMU_EPI_RESID) && !resid)) return -22;
    F8Args a = {(const unsigned char*)A, (const unsigned char*)SA, (const unsigned char*)B, (const unsigned char*)SB, (bf16*)C,
                bias, (const bf16*)resid, lda, ldsa, ldb, ldsb, ldc, ldr, M, N, K, flags, drop_seed,
                drop_threshold16(drop_p), drop_keep_scale16(drop_threshold16(drop_p))};
    if (!(flags & COMMU_EPI_DROPOUT) || drop_p <= 0.f) a.flags &= ~COMMU_EPI_DROPOUT;
    const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    COMMU_LAUNCH(gemm_nt_mxfp8_kernel, dim3(tiles), dim3(256), 0, stream, a);
    COMMU_LAUNCH_CHECK();
    return 0;
}

COMMU_DEFINE_SEED_SALT_SETTER(commu_seed_salt_gemm_fp8)
