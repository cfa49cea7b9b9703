// Internal interface between gemm.hip (dispatch) and gemm8.hip (the 256 x 256 x 64 eight-phase kernels).
#pragma once
#include "common.h"

struct G8Args {
    const bf16* A;
    const bf16* B;
    void* C;
    int lda, ldb, ldc, M, N, K, tiles_m, tiles_n;
    const float* bias;
    const bf16* resid;
    int ldr;
    const bf16* rmask;
    int ldm, flags;
    unsigned drop_seed, drop_thr;
    float drop_scale, mask_scale;
    int store_mode;           // (experiment) 0 plain, 1 nt, 2 sc1, 3 sc0 sc1
    int skew_cycles;          // start-up skew between the workgroups of an XCD (0: none)
};

// true when the 8-phase NT kernel takes this problem (large M, K % 64 == 0, 32-bit buffer offsets)
bool gemm8_nt_eligible(int M, int N, int K, int lda, int ldb, int batch, int tri_B, int flags);
int launch_gemm8_nt(const G8Args& a, hipStream_t stream);
// sign-bit epilogues (COMMU_EPI_SIGNBITS_OUT / COMMU_EPI_RELUBITS; G8Args.rmask is the word buffer): every wave interior
bool gemm8_nt_bits_eligible(int M, int N, int K, int lda, int ldb, int ldc, int flags);

// ---- grouped TN (weight gradients): out_p[n, k] = sum_m A_p[m, n] * B_p[m, k] for up to 8 problems that share M
struct Tn8Prob {
    const bf16* A;          // [M, N] (dY), row stride lda
    const bf16* B;          // [M, Kc] (layer input), row stride ldb
    int lda, ldb, N, Kc;
    int tiles_n, tiles_k, tile0;          // 256 x 256 output tiles; first tile id of this problem
    long long out_off;                    // element offset of this problem's [N, Kc] block inside a slab
    long long colsum_off;                 // >= 0: column sums of A ([N] floats) at this slab offset
};
struct Tn8Args {
    Tn8Prob p[8];
    int nprob, M, nslices, m_per_slice, total_tiles;
    float* slabs;               // [nslices][slab_stride]
    long long slab_stride;
};
bool gemm8_tn_eligible(int M, int N, int Kc, int lda, int ldb);
int launch_gemm8_tn(const Tn8Args& a, hipStream_t stream);
