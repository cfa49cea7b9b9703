// bf16 MFMA GEMMs for the Transformer-XL hot path (gfx950).
//
//  gemm_nt : C[M,N] = A[M,K] . B[N,K]^T (+ fused epilogue)       -- every forward Linear
//            (reference: nn.Linear calls at commu/model/model.py:205,212,278,164,167,46) and
//            every dX = dY . W backward (W passed pre-transposed, so it is NT again).
//  gemm_tn : C[N,K] = sum_m A[m,N]^T . B[m,K]   (split over m)     -- every dW = dY^T . X.
//
// Tiling: 128x128 output tile, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA
// 16x16x32 blocks, K step 32, register-staged double-buffered LDS (one barrier per K step).
// The MFMA is issued "swapped" (weight rows as the A operand) so that each lane ends up with
// 4 consecutive output columns of one row: 8-byte bf16x4 / 16-byte f32x4 stores.
#include "common.h"
#include "commu_hip.h"
#include "gemm8.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int BK = 32;

// 64-byte LDS rows (32 bf16): XOR the 16-byte chunk index so that each ds_read_b128 lane
// group ({0-3,12-15,20-27}, ...) touches 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int swz64(int row) { return ((row >> 3) & 1) * 3; }

struct GemmBatch {          // element strides between consecutive batch entries (blockIdx.y)
    long long a, b, c, r;
    int tri_B, tri_M;       // causal band: row m = i*tri_B + b of A is zero beyond column i + tri_M (0: off)
};

// NBW = 16-column blocks per wave: 4 -> 128-wide tile, 2 -> 64-wide tile (per-head GEMMs, N = d_head)
// swizzle of the 16-byte chunk index for a row of BKT bf16 (64 or 128 bytes)
template <int BKT>
__device__ __forceinline__ int swzk(int row) { return BKT == 64 ? (row & 7) : swz64(row); }

// Which weight row (= output column) sits at LDS row r of the B tile.  Within a wave's 16*NBW columns the
// MFMA output gives lane (r16, g) the elements (block ni, 4g + reg); staging the columns in this order makes
// those 4*NBW/... elements 8 CONSECUTIVE output columns per pair of blocks: column = 32*(ni>>1) + 8g + 4*(ni&1)
// + reg, so the epilogue stores 16 bytes per lane and 64 contiguous bytes per row and instruction.
template <int NBW>
__device__ __forceinline__ int colperm(int r) {
    const int c = r & 15, ni = (r >> 4) & (NBW - 1);
    if (NBW == 4) return (r & ~63) + 32 * (ni >> 1) + 8 * (c >> 2) + 4 * (ni & 1) + (c & 3);
    return (r & ~31) + 8 * (c >> 2) + 4 * ni + (c & 3);
}

// Tile = (16*MBW*WM) x (16*NBW*WN): WM x WN waves, each MBW x NBW MFMA blocks.  Configurations:
//   <4,4|2,2,2>: 128 x 128|64, 256 threads     <8,4,2,4>: 256 x 256, 512 threads (halves the bytes each
//   CU loads per flop: the 128x128 tile is bound by the ~15 B/cycle/CU vector-load path, not by MFMA)
// GLDS: tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write); the LDS image
// is lane-linear, so the XOR swizzle is applied to the per-lane SOURCE address instead.
template <bool OUT_F32, int NBW, int BKT, int MBW = 4, int WM = 2, int WN = 2, bool GLDS = true>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(
    const bf16* __restrict__ A, int lda, const bf16* __restrict__ B, int ldb,
    void* __restrict__ Cv, int ldc, int M, int N, int K,
    const float* __restrict__ bias, const bf16* __restrict__ resid, int ldr,
    const bf16* __restrict__ rmask, int ldm, int flags, int tiles_n, unsigned drop_seed, unsigned drop_thr,
    float drop_scale, float mask_scale, GemmBatch bs) {
    constexpr int BM = 16 * MBW * WM;
    constexpr int BN = 16 * NBW * WN;
    constexpr int BK = BKT;
    constexpr int NTHR = 64 * WM * WN;
    constexpr int CPR = BK / 8;                 // 16-byte chunks per tile row (4 or 8)
    constexpr int RPP = NTHR / CPR;             // tile rows covered by one pass of all threads
    __shared__ __attribute__((aligned(16))) bf16 sA[2][BM * BK];
    __shared__ __attribute__((aligned(16))) bf16 sB[2][BN * BK];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w / WN, wc = w % WN, r16 = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int tm = lid / tiles_n;
    const int tn = lid - tm * tiles_n;
    if (bs.tri_B > 0) tm = (int)gridDim.x / tiles_n - 1 - tm;      // causal band: longest contractions (last rows) first
    const int m0 = tm * BM, n0 = tn * BN;
    A += (long long)blockIdx.y * bs.a;
    B += (long long)blockIdx.y * bs.b;
    if (resid != nullptr) resid += (long long)blockIdx.y * bs.r;
    const long long coff = (long long)blockIdx.y * bs.c;

    const int lrow = tid / CPR, lch = tid % CPR;
    constexpr int NLA = BM / RPP;         // A-tile chunks per thread
    constexpr int NLB = BN / RPP;         // B-tile chunks per thread
    const bf16* ap[NLA];
    const bf16* bp[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) ap[i] = A + (size_t)min(m0 + lrow + RPP * i, M - 1) * lda + lch * 8;
#pragma unroll
    for (int i = 0; i < NLB; ++i) bp[i] = B + (size_t)min(n0 + colperm<NBW>(lrow + RPP * i), N - 1) * ldb + lch * 8;
    bf16x8 ra[NLA], rb[NLB];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) ra[i] = ld_bf16x8(ap[i] + kt * BK);
#pragma unroll
        for (int i = 0; i < NLB; ++i) rb[i] = ld_bf16x8(bp[i] + kt * BK);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int r = lrow + RPP * i;
            st_bf16x8(&sA[buf][r * BK + ((lch ^ swzk<BKT>(r)) << 3)], ra[i]);
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int r = lrow + RPP * i;
            st_bf16x8(&sB[buf][r * BK + ((lch ^ swzk<BKT>(r)) << 3)], rb[i]);
        }
    };

    f32x4 acc[NBW][MBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int j = 0; j < MBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int keff = K;
    if (bs.tri_B > 0) keff = min(K, ((min(m0 + BM, M) - 1) / bs.tri_B + bs.tri_M + 1 + 63) & ~63);
    const int nk = keff / BK;
    // ---- LDS-DMA staging: chunk q = tid + NTHR*i of a tile lands at LDS byte 16*q; it holds tile row q / CPR,
    // logical chunk (q % CPR) ^ swz(row)
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    const bf16* gpa[NLA];
    const bf16* gpb[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int q = tid + NTHR * i, r = q / CPR, pc = q % CPR;
        gpa[i] = A + (size_t)min(m0 + r, M - 1) * lda + ((pc ^ swzk<BKT>(r)) << 3);
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int q = tid + NTHR * i, r = q / CPR, pc = q % CPR;
        gpb[i] = B + (size_t)min(n0 + colperm<NBW>(r), N - 1) * ldb + ((pc ^ swzk<BKT>(r)) << 3);
    }
    auto glds = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < NLA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gpa[i] + kt * BK),
                                             (LDS_AS void*)(&sA[buf][(wbase + NTHR * i) * 8]), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NLB; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gpb[i] + kt * BK),
                                             (LDS_AS void*)(&sB[buf][(wbase + NTHR * i) * 8]), 16, 0, 0);
    };
    if (GLDS) {
        glds(0, 0);
    } else {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (!GLDS && kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            // (all fragment rows are r16 + a multiple of 16, so the swizzle term depends on r16 only)
            const int choff = ((4 * ks + g) ^ swzk<BKT>(r16)) << 3;
            bf16x8 af[MBW], bfr[NBW];
#pragma unroll
            for (int i = 0; i < MBW; ++i) af[i] = ld_bf16x8(&sA[buf][(wr * 16 * MBW + 16 * i + r16) * BK + choff]);
#pragma unroll
            for (int i = 0; i < NBW; ++i) bfr[i] = ld_bf16x8(&sB[buf][(wc * 16 * NBW + 16 * i + r16) * BK + choff]);
            if (GLDS && ks == BK / 32 - 1) {
                // the LDS-DMA of the next tile goes out AFTER this tile's last fragment reads: the compiler puts
                // a vmcnt(0) in front of the first LDS read that follows a DMA, so issuing it earlier would
                // serialise load -> wait -> MFMA.  Here it flies under the MFMAs below.
                __builtin_amdgcn_sched_barrier(0);
                if (kt + 1 < nk) glds(kt + 1, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ni = 0; ni < NBW; ++ni)
#pragma unroll
                for (int mi = 0; mi < MBW; ++mi) acc[ni][mi] = mfma16(bfr[ni], af[mi], acc[ni][mi]);
        }
        if (!GLDS && kt + 1 < nk) lstore(buf ^ 1);
        if (GLDS) {
            __builtin_amdgcn_sched_barrier(0);                      // keep the MFMAs above the wait
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // my DMA pieces have landed before the barrier
        }
        __syncthreads();
    }

    // epilogue: lane holds C[m = .. + r16][n = .. + 32*j + 8g + e], e = 4*(ni&1) + reg, ni = 2j + (e>>2)
#pragma unroll
    for (int mi = 0; mi < MBW; ++mi) {
        const int m = m0 + wr * 16 * MBW + 16 * mi + r16;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NBW / 2; ++j) {
            const int n = n0 + wc * 16 * NBW + (NBW == 4 ? 32 * j : 0) + 8 * g;
            if (n >= N) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[2 * j + (e >> 2)][mi][e & 3];
            const bool full = (n + 7 < N);
            if (flags & COMMU_EPI_BIAS) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < N) v[e] += bias[n + e];
            }
            if (flags & COMMU_EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (flags & COMMU_EPI_DROPOUT) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    v[e] = drop_keep(salted(drop_seed), (unsigned)m * (unsigned)N + (unsigned)(n + e), drop_thr) ? v[e] * drop_scale : 0.f;
            }
            if (flags & COMMU_EPI_RESID) {
                const bf16* rp = resid + (size_t)m * ldr + n;
                if (full && (ldr % 8) == 0) {
                    const bf16x8 r8 = ld_bf16x8(rp);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf2f(r8[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < N) v[e] += bf2f(rp[e]);
                }
            }
            if (flags & COMMU_EPI_RELUMASK) {
                const bf16* mp = rmask + (size_t)m * ldm + n;
                if (full && (ldm % 8) == 0) {
                    const bf16x8 m8 = ld_bf16x8(mp);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf2f(m8[e]) > 0.f) ? v[e] * mask_scale : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < N) v[e] = (bf2f(mp[e]) > 0.f) ? v[e] * mask_scale : 0.f;
                }
            }
            if (OUT_F32) {
                float* C = (float*)Cv + coff + (size_t)m * ldc + n;
                if (full) {
                    *(f32x4*)C = (f32x4){v[0], v[1], v[2], v[3]};
                    *(f32x4*)(C + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                } else {
                    for (int e = 0; e < 8; ++e)
                        if (n + e < N) C[e] = v[e];
                }
            } else {
                bf16* C = (bf16*)Cv + coff + (size_t)m * ldc + n;
                if (full && (ldc % 8) == 0) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
                    st_bf16x8(C, o);
                } else {
                    for (int e = 0; e < 8; ++e)
                        if (n + e < N) C[e] = f2bf(v[e]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Skinny NT GEMM for the decode step (M <= 64 rows: one token per sequence): C[M, N] = A[M, K] . B[N, K]^T.
// The work is a single pass over the weights, so the kernel is latency-, not throughput-bound: the tiled
// kernel's 2-stage pipeline walks K in 8-16 dependent HBM round trips (17 us per GEMM).  Here a workgroup owns
// 32 output columns, its 4 waves split K, and every wave issues ALL its fragment loads (weights and
// activations, straight from global memory in MFMA operand layout) before the first MFMA: one round trip.
// Partial sums meet in LDS; epilogue bias -> ReLU -> residual.
// LN = true: A is the PRE-LayerNorm activation z; the kernel normalises it on the fly (a = LN(z) over the first D
// columns, gamma / beta / eps: nn.LayerNorm at model.py:179,352) and workgroup 0 also stores a (bf16) to a_out, where the
// next residual add reads it.  Every workgroup holds all M <= 64 rows anyway, so the row statistics cost two tiny LDS
// reductions and the decode step loses one kernel launch and one activation round trip per LayerNorm.
struct SkinnyLN {
    const float* gamma;
    const float* beta;
    bf16* a_out;
    int lda_out, D;
    float eps;
};

template <int KSTEPS, bool OUT_F32, int NWV = 4, bool LN = false>          // KSTEPS = K / (32 NWV): 32-wide MFMA steps per wave
__global__ __launch_bounds__(64 * NWV) void gemm_nt_skinny_kernel(
    const bf16* __restrict__ A, int lda, const bf16* __restrict__ B, int ldb, void* __restrict__ Cv, int ldc,
    int M, int N, const float* __restrict__ bias, const bf16* __restrict__ resid, int ldr, int flags, SkinnyLN ln) {
    __shared__ float red[NWV][32][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 32;
    const int k0 = w * KSTEPS * 32 + 8 * g;
    bf16x8 bfr[2][KSTEPS], af[4][KSTEPS];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const bf16* bp = B + (size_t)min(n0 + 16 * ni + r16, N - 1) * ldb + k0;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) bfr[ni][ks] = ld_bf16x8(bp + 32 * ks);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const bf16* ap = A + (size_t)min(16 * mi + r16, M - 1) * lda + k0;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) af[mi][ks] = ld_bf16x8(ap + 32 * ks);
    }
    // (LN) gamma / beta of this wave's K range: requested together with the fragments, not after the statistics
    float gm[LN ? KSTEPS : 1][8], bt[LN ? KSTEPS : 1][8];
    if (LN) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = k0 + 32 * ks + e;
                gm[ks][e] = kk < ln.D ? ln.gamma[kk] : 0.f;
                bt[ks][e] = kk < ln.D ? ln.beta[kk] : 0.f;
            }
    }
    __builtin_amdgcn_sched_barrier(0);          // every load above is in flight before the first MFMA waits
    if (LN) {
        // row statistics: lane (r16, g) holds 8 KSTEPS values of rows 16 mi + r16 -> reduce over g (lanes 16 apart),
        // then over the waves through LDS.  Two passes (mean, then centred squares), like layernorm_fwd_kernel.
        float* st = &red[0][0][0];          // [NWV][64 rows]
        const int D = ln.D;
        float mu[4], rs[4];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                float v = 0.f;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = (k0 + 32 * ks + e < D) ? bf2f(af[mi][ks][e]) : 0.f;
                        v += pass == 0 ? x : ((k0 + 32 * ks + e < D) ? (x - mu[mi]) * (x - mu[mi]) : 0.f);
                    }
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (g == 0) st[w * 64 + 16 * mi + r16] = v;
            }
            __syncthreads();
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                float t = 0.f;
#pragma unroll
                for (int ww = 0; ww < NWV; ++ww) t += st[ww * 64 + 16 * mi + r16];
                if (pass == 0) mu[mi] = t / (float)D;
                else rs[mi] = rsqrtf(t / (float)D + ln.eps);
            }
            __syncthreads();
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int kk = k0 + 32 * ks;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    af[mi][ks][e] = f2bf(kk + e < D ? (bf2f(af[mi][ks][e]) - mu[mi]) * rs[mi] * gm[ks][e] + bt[ks][e] : 0.f);
                if (blockIdx.x == 0 && ln.a_out != nullptr && 16 * mi + r16 < M)
                    st_bf16x8(ln.a_out + (size_t)(16 * mi + r16) * ln.lda_out + kk, af[mi][ks]);
            }
        }
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16(bfr[ni][ks], af[mi][ks], acc[ni][mi]);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[w][(ni * 4 + mi) * 4 + e][lane] = acc[ni][mi][e];
    __syncthreads();
    // the 8 (ni, mi) blocks are shared out over the waves; lane holds C[16 mi + r16][n0 + 16 ni + 4 g + e]
    constexpr int BPW = 8 / NWV;
#pragma unroll
    for (int q = 0; q < BPW; ++q) {
        const int blk = w * BPW + q, ni = blk >> 2, mi = blk & 3;
        const int m = 16 * mi + r16, n = n0 + 16 * ni + 4 * g;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = (ni * 4 + mi) * 4 + e;
            float acc1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < NWV; ++ww) acc1 += red[ww][idx][lane];
            v[e] = acc1;
        }
        if (m >= M || n >= N) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n + e < N) {
                if (flags & COMMU_EPI_BIAS) v[e] += bias[n + e];
                if (flags & COMMU_EPI_RELU) v[e] = fmaxf(v[e], 0.f);
                if (flags & COMMU_EPI_RESID) v[e] += bf2f(resid[(size_t)m * ldr + n + e]);
            }
        }
        if (OUT_F32) {
            float* C = (float*)Cv + (size_t)m * ldc + n;
            if (n + 3 < N) *(f32x4*)C = (f32x4){v[0], v[1], v[2], v[3]};
            else for (int e = 0; e < 4; ++e) if (n + e < N) C[e] = v[e];
        } else {
            bf16* C = (bf16*)Cv + (size_t)m * ldc + n;
            if (n + 3 < N) *(bf16x4*)C = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            else for (int e = 0; e < 4; ++e) if (n + e < N) C[e] = f2bf(v[e]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// TN: out[n, k] (fp32 slab per m-slice) = sum_{m in slice} A[m, n] * B[m, k].
// Both operands have the contraction index m as their ROW index, so MFMA fragments (8
// consecutive m per lane) are columns of the staged [32 m][128] LDS images.
//   mode 1: ds_read_b64_tr_b16 transpose reads (2 per fragment)
//   mode 0: ds_read_u16 gathers (8 per fragment) -- slow, correct by construction.
constexpr int TM = 64;      // contraction rows per pipeline step (two MFMA k-steps)

// swizzle of the 32-byte unit index (8 units per 256-byte row... 128 cols) used by the TN images
__device__ __forceinline__ int swz_tn(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

typedef __amdgpu_buffer_rsrc_t tn_srd_t;

// Tile BNA (n) x BNB (k), WR x WC waves; TMS contraction rows per pipeline step.  Configurations:
//   <128,128,2,2,64>, <128,64,2,2,64> (per-head outputs, K <= 64): 256 threads
//   <256,256,2,4,32>: 512 threads, wave tile 128 x 64 -- halves the bytes each CU pulls through its vector-load
//   path per flop (the 128 x 128 tile runs at ~13 B/cycle/CU of L2->LDS traffic, its limit)
template <int BNA, int BNB, int WR, int WC, int TMS, int MODE>
__global__ __launch_bounds__(64 * WR * WC) void gemm_tn_kernel(
    const bf16* __restrict__ A, int lda, const bf16* __restrict__ B, int ldb,
    float* __restrict__ C, int ldc, size_t slab_stride, int Mtot, int N, int Kc, int m_per_slice, int nslices,
    long long strideA, long long strideB, int tri_B, int tri_M) {
    constexpr int NTHR = 64 * WR * WC;
    __shared__ __attribute__((aligned(16))) bf16 sA[2][TMS * BNA];      // [m][n] images, double buffered
    __shared__ __attribute__((aligned(16))) bf16 sB[2][TMS * BNB];
    constexpr int WN = BNA / WR, WK = BNB / WC;      // wave tile
    constexpr int NB = WN / 16, KB = WK / 16;        // MFMA blocks per wave
    constexpr int SWA = (BNA / 16 - 1) & 7, SWB = (BNB / 16 - 1) & 7;      // swizzle masks (32-byte units per row - 1, <= 7)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w / WC, wc = w % WC, r16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * BNA, k0 = blockIdx.y * BNB;
    const int batch = blockIdx.z / nslices;
    const int slice = blockIdx.z - batch * nslices;
    A += (long long)batch * strideA;
    B += (long long)batch * strideB;
    int mbeg = slice * m_per_slice;
    const int mend = min(Mtot, mbeg + m_per_slice);
    if (tri_B > 0) mbeg = max(mbeg, (max(n0 - tri_M, 0) * tri_B) & ~(TMS - 1));      // rows above the band are zero
    const int nsteps = max(0, (mend - mbeg + TMS - 1) / TMS);

    // staging with buffer loads: rows >= mend (and the bytes past column N / Kc of the last row) lie outside the
    // SRD's range and read as zero -- no predicates.  Tile columns >= N (or Kc) are forced out of range.
    const tn_srd_t srdA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0,
        mend > 0 ? (int)(unsigned)((((size_t)(mend - 1)) * lda + N) * 2) : 0, 0x00020000);
    const tn_srd_t srdB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0,
        mend > 0 ? (int)(unsigned)((((size_t)(mend - 1)) * ldb + Kc) * 2) : 0, 0x00020000);
    constexpr int CHA = BNA / 8, CHB = BNB / 8;              // 16-byte chunks per image row
    constexpr int RPA = NTHR / CHA, RPB = NTHR / CHB;        // rows per pass of all threads
    constexpr int NLA = TMS / RPA, NLB = TMS / RPB;
    static_assert(NLA >= 1 && NLB >= 1 && TMS % RPA == 0 && TMS % RPB == 0, "staging shape");
    const int arow = tid / CHA, ach = tid % CHA, brow = tid / CHB, bch = tid % CHB;
    bf16x8 ra[NLA], rb[NLB];
    const unsigned aoff = ((unsigned)arow * lda + n0 + ach * 8) * 2u, boff = ((unsigned)brow * ldb + k0 + bch * 8) * 2u;
    const bool acol = n0 + ach * 8 < N, bcol = k0 + bch * 8 < Kc;      // (whole 16-byte chunks: N, Kc % 8 == 0)
    auto gload = [&](int st) {
        const unsigned mb = (unsigned)(mbeg + st * TMS);
#pragma unroll
        for (int i = 0; i < NLA; ++i)
            ra[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                srdA, acol ? (int)(aoff + (mb + RPA * i) * (unsigned)lda * 2u) : -16, 0, 0));
#pragma unroll
        for (int i = 0; i < NLB; ++i)
            rb[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                srdB, bcol ? (int)(boff + (mb + RPB * i) * (unsigned)ldb * 2u) : -16, 0, 0));
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int r = arow + RPA * i;
            const int ch = (((ach >> 1) ^ (swz_tn(r) & SWA)) << 1) | (ach & 1);
            st_bf16x8(&sA[buf][r * BNA + ch * 8], ra[i]);
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int r = brow + RPB * i;
            const int ch = (((bch >> 1) ^ (swz_tn(r) & SWB)) << 1) | (bch & 1);
            st_bf16x8(&sB[buf][r * BNB + ch * 8], rb[i]);
        }
    };
    // fragment = 8 consecutive m (8g .. 8g+7) of column `col` of a 32-row image with `ncols` columns
    auto frag = [&](const bf16* img, int ncols, int swm, int colbase) -> bf16x8 {
        bf16x8 f;
        if (MODE == 1) {
            // lane i of a 16-lane group supplies &img[8g + (i>>2) (+4)][colbase + 4*(i&3)]
            const int i = r16;
            const int c = colbase + 4 * (i & 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 8 * g + (i >> 2) + 4 * h;
                const int unit = (c >> 4) ^ (swz_tn(r) & swm);
                const bf16* p = img + r * ncols + unit * 16 + (c & 15);
                typedef __attribute__((ext_vector_type(4))) short s16x4;
                s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p));
                bf16x4 vb = __builtin_bit_cast(bf16x4, v);
                f[4 * h + 0] = vb[0]; f[4 * h + 1] = vb[1]; f[4 * h + 2] = vb[2]; f[4 * h + 3] = vb[3];
            }
        } else {
            const int c = colbase + r16;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * g + e;
                const int unit = (c >> 4) ^ (swz_tn(r) & swm);
                f[e] = img[r * ncols + unit * 16 + (c & 15)];
            }
        }
        return f;
    };

    f32x4 acc[KB][NB];
#pragma unroll
    for (int i = 0; i < KB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int buf = st & 1;
        if (st + 1 < nsteps) gload(st + 1);
#pragma unroll
        for (int ks = 0; ks < TMS / 32; ++ks) {
            bf16x8 af[NB], bfr[KB];
#pragma unroll
            for (int i = 0; i < NB; ++i) af[i] = frag(sA[buf] + 32 * ks * BNA, BNA, SWA, wr * WN + 16 * i);
#pragma unroll
            for (int i = 0; i < KB; ++i) bfr[i] = frag(sB[buf] + 32 * ks * BNB, BNB, SWB, wc * WK + 16 * i);
            // swapped issue: D[k][n] -> lane holds out[n = .. + r16][k = .. + 4g + reg]
#pragma unroll
            for (int ki = 0; ki < KB; ++ki)
#pragma unroll
                for (int ni = 0; ni < NB; ++ni) acc[ki][ni] = mfma16(bfr[ki], af[ni], acc[ki][ni]);
        }
        if (st + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }
    float* Cs = C + (size_t)blockIdx.z * slab_stride;
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) {
        const int n = n0 + wr * WN + 16 * ni + r16;
        if (n >= N) continue;
#pragma unroll
        for (int ki = 0; ki < KB; ++ki) {
            const int k = k0 + wc * WK + 16 * ki + 4 * g;
            if (k + 3 < Kc) {
                *(f32x4*)(Cs + (size_t)n * ldc + k) = acc[ki][ni];
            } else {
                for (int e = 0; e < 4; ++e)
                    if (k + e < Kc) Cs[(size_t)n * ldc + k + e] = acc[ki][ni][e];
            }
        }
    }
}

// dst[i] = (accumulate ? dst[i] : 0) + alpha * sum_s src[s*stride + i]
__global__ void reduce_slabs_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n,
                                    int nslabs, size_t stride, int accumulate, float alpha) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t step = (size_t)gridDim.x * blockDim.x * 4;
    for (; i < n; i += step) {
        if (i + 3 < n) {
            f32x4 s = *(const f32x4*)(src + i);
            for (int k = 1; k < nslabs; ++k) s += *(const f32x4*)(src + (size_t)k * stride + i);
            s *= alpha;
            if (accumulate) s += *(const f32x4*)(dst + i);
            *(f32x4*)(dst + i) = s;
        } else {
            for (size_t j = i; j < n; ++j) {
                float s = src[j];
                for (int k = 1; k < nslabs; ++k) s += src[(size_t)k * stride + j];
                s *= alpha;
                if (accumulate) s += dst[j];
                dst[j] = s;
            }
        }
    }
}

// dst[z][r*ldd + c] = (accumulate ? dst : 0) + alpha * sum_s src[(z*nslabs + s)*stride + r*cols + c]
__global__ void reduce_slabs2d_kernel(float* __restrict__ dst, int ldd, long long dst_bs, const float* __restrict__ src,
                                      int rows, int cols, int nslabs, size_t stride, int accumulate, float alpha) {
    const int z = blockIdx.y;
    const size_t n = (size_t)rows * cols;
    const float* sp = src + (size_t)z * nslabs * stride;
    float* dp = dst + (long long)z * dst_bs;
    if ((cols & 3) == 0 && (stride & 3) == 0 && (ldd & 3) == 0 && (((size_t)sp | (size_t)dp) & 15) == 0) {
        // four elements per thread, eight slabs in flight (scalar loads one slab at a time: 94 us for the 64 MB of a layer's
        // dRd slabs, 0.7 TB/s)
        for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            int k = 0;
            for (; k + 8 <= nslabs; k += 8) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(sp + (size_t)(k + u) * stride + i);
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; k < nslabs; ++k) s += *(const f32x4*)(sp + (size_t)k * stride + i);
            const int r = (int)((unsigned)i / (unsigned)cols), c = (int)((unsigned)i - (unsigned)r * (unsigned)cols);
            f32x4* o = (f32x4*)(dp + (size_t)r * ldd + c);
            *o = (accumulate ? *o : (f32x4){0.f, 0.f, 0.f, 0.f}) + alpha * s;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nslabs; ++k) s += sp[(size_t)k * stride + i];
        const int r = (int)((unsigned)i / (unsigned)cols), c = (int)((unsigned)i - (unsigned)r * (unsigned)cols);   // (n < 2^32)
        float* o = dp + (size_t)r * ldd + c;
        *o = (accumulate ? *o : 0.f) + alpha * s;
    }
}

}  // namespace

static int launch_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                          const float* bias, const void* resid, int ldr, const void* relu_mask, int ldm, int flags,
                          unsigned drop_seed, float drop_p, float mask_scale, int batch, GemmBatch bs,
                          hipStream_t stream, const SkinnyLN* ln = nullptr) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (K <= 0 || (K % 32) != 0 || (lda % 8) || (ldb % 8) || (ldc % 4)) return -22;
    if (flags & (COMMU_EPI_SIGNBITS_OUT | COMMU_EPI_RELUBITS)) {
        // one bit per output instead of the bf16 ReLU mask: the eight-phase kernel's pipelined epilogue only
        if (batch != 1 || ln != nullptr || relu_mask == nullptr || !gemm8_nt_bits_eligible(M, N, K, lda, ldb, ldc, flags))
            return -22;
        G8Args g8{(const bf16*)A, (const bf16*)B, C, lda, ldb, ldc, M, N, K, (M + 255) / 256, (N + 255) / 256,
                  bias, nullptr, 0, (const bf16*)relu_mask, 0, flags, drop_seed,
                  drop_threshold16(drop_p), drop_keep_scale16(drop_threshold16(drop_p)), mask_scale, 0, 0};
        return launch_gemm8_nt(g8, stream);
    }
    const bool skinny = M <= 64 && batch == 1 && bs.tri_B == 0 && (K % 128) == 0 && K <= 1024 && N >= 32 &&
                        !(flags & (COMMU_EPI_DROPOUT | COMMU_EPI_RELUMASK));
    if (ln != nullptr && !skinny) return -22;          // the LayerNorm-fused form exists for the decode step only
    if (skinny && (ln != nullptr || !getenv("COMMU_GEMM_NOSKINNY"))) {
        dim3 grid((N + 31) / 32);
#define SK_LAUNCH(KS, NWV)                                                                                           \
    {                                                                                                                \
        if (ln != nullptr) {                                                                                         \
            if (flags & COMMU_EPI_OUT_F32)                                                                           \
                COMMU_LAUNCH((gemm_nt_skinny_kernel<KS, true, NWV, true>), grid, dim3(64 * NWV), 0, stream,          \
                             (const bf16*)A, lda, (const bf16*)B, ldb, C, ldc, M, N, bias, (const bf16*)resid, ldr,  \
                             flags, *ln);                                                                            \
            else                                                                                                     \
                COMMU_LAUNCH((gemm_nt_skinny_kernel<KS, false, NWV, true>), grid, dim3(64 * NWV), 0, stream,         \
                             (const bf16*)A, lda, (const bf16*)B, ldb, C, ldc, M, N, bias, (const bf16*)resid, ldr,  \
                             flags, *ln);                                                                            \
        } else if (flags & COMMU_EPI_OUT_F32)                                                                        \
            COMMU_LAUNCH((gemm_nt_skinny_kernel<KS, true, NWV>), grid, dim3(64 * NWV), 0, stream, (const bf16*)A,    \
                         lda, (const bf16*)B, ldb, C, ldc, M, N, bias, (const bf16*)resid, ldr, flags, SkinnyLN{});  \
        else                                                                                                         \
            COMMU_LAUNCH((gemm_nt_skinny_kernel<KS, false, NWV>), grid, dim3(64 * NWV), 0, stream, (const bf16*)A,   \
                         lda, (const bf16*)B, ldb, C, ldc, M, N, bias, (const bf16*)resid, ldr, flags, SkinnyLN{});  \
    }
        if (K % 256 == 0 && K >= 768) {          // long contraction: 8 waves split K
            switch (K / 256) {
                case 3: SK_LAUNCH(3, 8) break;
                default: SK_LAUNCH(4, 8) break;
            }
        } else {
            switch (K / 128) {
                case 1: SK_LAUNCH(1, 4) break;
                case 2: SK_LAUNCH(2, 4) break;
                case 3: SK_LAUNCH(3, 4) break;
                case 4: SK_LAUNCH(4, 4) break;
                case 5: SK_LAUNCH(5, 4) break;
                case 6: SK_LAUNCH(6, 4) break;
                case 7: SK_LAUNCH(7, 4) break;
                default: SK_LAUNCH(8, 4) break;
            }
        }
#undef SK_LAUNCH
        COMMU_LAUNCH_CHECK();
        return 0;
    }
    if (gemm8_nt_eligible(M, N, K, lda, ldb, batch, bs.tri_B, flags)) {
        // large-M Linear shapes: persistent 256 x 256 x 64 eight-phase kernel (gemm8.hip)
        G8Args g8{(const bf16*)A, (const bf16*)B, C, lda, ldb, ldc, M, N, K, (M + 255) / 256, (N + 255) / 256,
                  bias, (const bf16*)resid, ldr, (const bf16*)relu_mask, ldm, flags, drop_seed,
                  drop_threshold16(drop_p), drop_keep_scale16(drop_threshold16(drop_p)), mask_scale, 0, 0};
        return launch_gemm8_nt(g8, stream);
    }
    const bool narrow = (N <= 64);
    // large M: 256 x 256 (512 threads) for wide outputs, 256 x 128 (256 threads, two workgroups per CU so one's
    // epilogue overlaps the other's main loop) otherwise; 128 x 128 for small problems
    const bool large = !narrow && M >= 2048 && N >= 128 && !getenv("COMMU_GEMM_SMALL");
    const bool big = large && N >= 1024 && (N % 256 == 0) && !getenv("COMMU_GEMM_TALL");
    const bool tall = large && !big;
    const int bn = narrow ? 64 : (tall ? 128 : (big ? 256 : 128)), bm = (big || tall) ? 256 : 128;
    const int tiles_m = (M + bm - 1) / bm, tiles_n = (N + bn - 1) / bn;
    dim3 grid(tiles_m * tiles_n, batch);
    const unsigned drop_thr = drop_threshold16(drop_p);          // 16-bit threshold, exact keep scale (common.h)
    const float drop_scale = drop_keep_scale16(drop_thr);
#define NT_LAUNCH(F32, NBW, BKT)                                                                                   \
    COMMU_LAUNCH((gemm_nt_kernel<F32, NBW, BKT>), grid, dim3(256), 0, stream, (const bf16*)A, lda, (const bf16*)B, \
                 ldb, C, ldc, M, N, K, bias, (const bf16*)resid, ldr, (const bf16*)relu_mask, ldm, flags,          \
                 tiles_n, drop_seed, drop_thr, drop_scale, mask_scale, bs)
    const bool k64 = (K % 64 == 0) && !getenv("COMMU_GEMM_BK32");
    if (tall) {
        if (flags & COMMU_EPI_OUT_F32)
            COMMU_LAUNCH((gemm_nt_kernel<true, 4, 32, 8, 2, 2>), grid, dim3(256), 0, stream, (const bf16*)A, lda,
                         (const bf16*)B, ldb, C, ldc, M, N, K, bias, (const bf16*)resid, ldr, (const bf16*)relu_mask,
                         ldm, flags, tiles_n, drop_seed, drop_thr, drop_scale, mask_scale, bs);
        else
            COMMU_LAUNCH((gemm_nt_kernel<false, 4, 32, 8, 2, 2>), grid, dim3(256), 0, stream, (const bf16*)A, lda,
                         (const bf16*)B, ldb, C, ldc, M, N, K, bias, (const bf16*)resid, ldr, (const bf16*)relu_mask,
                         ldm, flags, tiles_n, drop_seed, drop_thr, drop_scale, mask_scale, bs);
    } else if (big) {
        if (flags & COMMU_EPI_OUT_F32)
            COMMU_LAUNCH((gemm_nt_kernel<true, 4, 32, 8, 2, 4>), grid, dim3(512), 0, stream, (const bf16*)A, lda,
                         (const bf16*)B, ldb, C, ldc, M, N, K, bias, (const bf16*)resid, ldr, (const bf16*)relu_mask,
                         ldm, flags, tiles_n, drop_seed, drop_thr, drop_scale, mask_scale, bs);
        else
            COMMU_LAUNCH((gemm_nt_kernel<false, 4, 32, 8, 2, 4>), grid, dim3(512), 0, stream, (const bf16*)A, lda,
                         (const bf16*)B, ldb, C, ldc, M, N, K, bias, (const bf16*)resid, ldr, (const bf16*)relu_mask,
                         ldm, flags, tiles_n, drop_seed, drop_thr, drop_scale, mask_scale, bs);
    } else if (flags & COMMU_EPI_OUT_F32) {
        if (narrow) { if (k64) NT_LAUNCH(true, 2, 64); else NT_LAUNCH(true, 2, 32); }
        else { if (k64) NT_LAUNCH(true, 4, 64); else NT_LAUNCH(true, 4, 32); }
    } else {
        if (narrow) { if (k64) NT_LAUNCH(false, 2, 64); else NT_LAUNCH(false, 2, 32); }
        else { if (k64) NT_LAUNCH(false, 4, 64); else NT_LAUNCH(false, 4, 32); }
    }
#undef NT_LAUNCH
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" long long commu_gemm_nt_signbits_words(int M, int N, int K, int lda, int ldb, int ldc) {
    if (M <= 0 || N <= 0 || !gemm8_nt_bits_eligible(M, N, K, lda, ldb, ldc, 0)) return 0;
    return (long long)M * N / 32;
}

extern "C" int commu_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                                  int M, int N, int K, const float* bias, const void* resid, int ldr,
                                  const void* relu_mask, int ldm, int flags, unsigned drop_seed, float drop_p,
                                  float mask_scale, hipStream_t stream) {
    return launch_gemm_nt(A, lda, B, ldb, C, ldc, M, N, K, bias, resid, ldr, relu_mask, ldm, flags, drop_seed, drop_p,
                          mask_scale, 1, GemmBatch{0, 0, 0, 0, 0, 0}, stream);
}

extern "C" int commu_gemm_nt_bf16_batched(const void* A, int lda, long long strideA, const void* B, int ldb,
                                          long long strideB, void* C, int ldc, long long strideC, int M, int N,
                                          int K, const void* resid, int ldr, long long strideR, int flags,
                                          int batch, int tri_B, int tri_M, hipStream_t stream) {
    return launch_gemm_nt(A, lda, B, ldb, C, ldc, M, N, K, nullptr, resid, ldr, nullptr, 0, flags, 0u, 0.f, 1.f, batch,
                          GemmBatch{strideA, strideB, strideC, strideR, tri_B, tri_M}, stream);
}

// tile choice of the TN GEMM (shared by the launcher and commu_gemm_tn_slices)
static int tn_tile(int M, int N, int K) {          // 0: 128x64, 1: 128x128, 2: 256x256
    if (K <= 64) return 0;
    if (N >= 256 && K >= 256 && M >= 8192 && !getenv("COMMU_TN_SMALL")) return 2;
    return 1;
}

extern "C" int commu_gemm_tn_slices(int M, int N, int K) {
    const int t = tn_tile(M, N, K);
    int s;
    if (t == 2) {          // one 512-thread workgroup per CU: fill the 256 CUs once, never 1.03 times
        const int tiles = ((N + 255) / 256) * ((K + 255) / 256);
        s = 256 / tiles;
    } else {
        const int tiles = ((N + 127) / 128) * (t == 0 ? (K + 63) / 64 : (K + 127) / 128);
        s = (512 + tiles - 1) / tiles;
    }
    s = s < 64 ? s : 64;
    const int cap = (M + 255) / 256;
    s = s < cap ? s : cap;
    return s > 1 ? s : 1;
}

static int launch_gemm_tn(const void* A, int lda, const void* B, int ldb, float* slabs, int ldc, size_t slab_stride,
                          int M, int N, int K, int nslices, int mode, int batch, long long strideA, long long strideB,
                          int tri_B, int tri_M, hipStream_t stream) {
    if (N <= 0 || K <= 0 || nslices <= 0 || batch <= 0) return 0;
    if ((lda % 8) || (ldb % 8) || (ldc % 4) || (N % 8) || (K % 8)) return -22;
    if ((size_t)M * lda * 2 >= 0xFFFFFFF0ull || (size_t)M * ldb * 2 >= 0xFFFFFFF0ull) return -22;      // 32-bit buffer offsets
    const int t = tn_tile(M, N, K);
    int mps = (M + nslices - 1) / nslices;
    mps = ((mps + 63) / 64) * 64;
#define TN_LAUNCH(BNA, BNB, WR, WC, TMS, MD)                                                                   \
    COMMU_LAUNCH((gemm_tn_kernel<BNA, BNB, WR, WC, TMS, MD>), dim3((N + BNA - 1) / BNA, (K + BNB - 1) / BNB,   \
                 nslices * batch), dim3(64 * WR * WC), 0, stream, (const bf16*)A, lda, (const bf16*)B, ldb,    \
                 slabs, ldc, slab_stride, M, N, K, mps, nslices, strideA, strideB, tri_B, tri_M)
    if (t == 0) {
        if (mode) TN_LAUNCH(128, 64, 2, 2, 32, 1); else TN_LAUNCH(128, 64, 2, 2, 32, 0);
    } else if (t == 2 && mode) {
        TN_LAUNCH(256, 256, 2, 4, 32, 1);
    } else {
        if (mode) TN_LAUNCH(128, 128, 2, 2, 32, 1); else TN_LAUNCH(128, 128, 2, 2, 32, 0);
    }
#undef TN_LAUNCH
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, float* slabs,
                                  int ldc, size_t slab_stride, int M, int N, int K, int nslices,
                                  int mode, hipStream_t stream) {
    return launch_gemm_tn(A, lda, B, ldb, slabs, ldc, slab_stride, M, N, K, nslices, mode, 1, 0, 0, 0, 0, stream);
}

/* batch entry z: slabs[(z*nslices + s)][n,k] */
extern "C" int commu_gemm_tn_bf16_batched(const void* A, int lda, long long strideA, const void* B, int ldb,
                                          long long strideB, float* slabs, int ldc, size_t slab_stride, int M,
                                          int N, int K, int nslices, int batch, int tri_B, int tri_M,
                                          hipStream_t stream) {
    return launch_gemm_tn(A, lda, B, ldb, slabs, ldc, slab_stride, M, N, K, nslices, 1, batch, strideA, strideB, tri_B,
                          tri_M, stream);
}

extern "C" int commu_reduce_slabs_f32(float* dst, const float* src, size_t n, int nslabs,
                                      size_t stride, int accumulate, float alpha, hipStream_t stream) {
    if (n == 0) return 0;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    COMMU_LAUNCH(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, src, n,
                       nslabs, stride, accumulate, alpha);
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_reduce_slabs2d_f32(float* dst, int ldd, long long dst_batch_stride, const float* src, int rows,
                                        int cols, int nslabs, size_t stride, int batch, int accumulate, float alpha,
                                        hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || batch <= 0) return 0;
    if ((size_t)rows * cols >= 0xFFFFFFFFull) return -22;
    size_t blocks = ((size_t)rows * cols / 4 + 255) / 256;          // (four elements per thread on the vector path)
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    COMMU_LAUNCH(reduce_slabs2d_kernel, dim3((unsigned)blocks, batch), dim3(256), 0, stream, dst, ldd, dst_batch_stride,
                 src, rows, cols, nslabs, stride, accumulate, alpha);
    COMMU_LAUNCH_CHECK();
    return 0;
}

// Cropping reduce for zero-padded weight gradients (d_head 50 -> 64, d_model 500 -> 512): the product has the padded
// shape [rg*rp, cg*cp]; its [rt, ct] blocks go to dst [rg*rt, cg*ct]:
//   dst[((a*rt + r)*cg + c)*ct + k] (+)= alpha * sum_s src[s*stride + ((a*rp + r)*cg + c)*cp + k]
__global__ void reduce_slabs_crop_kernel(float* __restrict__ dst, const float* __restrict__ src, int rg, int rt, int rp,
                                         int cg, int ct, int cp, int nslabs, size_t stride, int accumulate, float alpha) {
    const size_t n = (size_t)rg * rt * cg * ct;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned q = (unsigned)i / (unsigned)ct;          // (n < 2^32: 32-bit divisions)
        const int k = (int)((unsigned)i - q * (unsigned)ct);
        unsigned q2 = q / (unsigned)cg;
        const int c = (int)(q - q2 * (unsigned)cg);
        const int a = (int)(q2 / (unsigned)rt), r = (int)(q2 - (unsigned)a * (unsigned)rt);
        const size_t j = (((size_t)a * rp + r) * cg + c) * cp + k;
        float v = src[j];
        for (int z = 1; z < nslabs; ++z) v += src[(size_t)z * stride + j];
        v *= alpha;
        if (accumulate) v += dst[i];
        dst[i] = v;
    }
}

extern "C" int commu_reduce_slabs_crop_f32(float* dst, const float* src, int rg, int rt, int rp, int cg, int ct, int cp,
                                           int nslabs, size_t stride, int accumulate, float alpha, hipStream_t stream) {
    if (rg <= 0 || rt <= 0 || cg <= 0 || ct <= 0) return 0;
    if (rt > rp || ct > cp || nslabs <= 0 || (size_t)rg * rt * cg * ct >= 0xFFFFFFFFull) return -22;
    size_t blocks = ((size_t)rg * rt * cg * ct + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    COMMU_LAUNCH(reduce_slabs_crop_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, src, rg, rt, rp, cg, ct, cp,
                 nslabs, stride, accumulate, alpha);
    COMMU_LAUNCH_CHECK();
    return 0;
}

// All slab reductions of a grouped weight-gradient launch in ONE launch: item z (blockIdx.y) is the cropping form above
// applied to its own block of the slabs and its own destination (a plain reduction is the crop (1, rows, rows_padded, 1,
// cols, cols)).
struct ReduceGroup {
    commu_reduce_item it[8];
};
__global__ void reduce_slabs_group_kernel(ReduceGroup grp, const float* __restrict__ slabs, int nslabs, size_t stride,
                                          int accumulate, float alpha) {
    const commu_reduce_item& it = grp.it[blockIdx.y];
    const float* src = slabs + it.src_off;
    float* dst = it.dst;
    const int rt = it.rt, rp = it.rp, cg = it.cg, ct = it.ct, cp = it.cp;
    const size_t n = (size_t)it.rg * rt * cg * ct;
    if (rt == rp && ct == cp) {
        // nothing to crop (the kernel-side shape IS the parameter's): source index == destination index -- no index
        // arithmetic, 16 bytes per access when aligned
        const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, tn = (size_t)gridDim.x * blockDim.x;
        if ((n & 3) == 0 && (stride & 3) == 0 && (((size_t)src | (size_t)dst) & 15) == 0) {
            for (size_t i = t0; i < (n >> 2); i += tn) {
                f32x4 v = ((const f32x4*)src)[i];
                for (int z = 1; z < nslabs; ++z) v += ((const f32x4*)(src + (size_t)z * stride))[i];
                v *= alpha;
                if (accumulate) v += ((const f32x4*)dst)[i];
                ((f32x4*)dst)[i] = v;
            }
        } else {
            for (size_t i = t0; i < n; i += tn) {
                float v = src[i];
                for (int z = 1; z < nslabs; ++z) v += src[(size_t)z * stride + i];
                v *= alpha;
                if (accumulate) v += dst[i];
                dst[i] = v;
            }
        }
        return;
    }
    // cropping: one destination row (rg * rt of them) per workgroup pass, one 32-bit division per element (the flat form
    // spent four 64-bit divisions per element: 55 us for a layer's 2.4 M gradients, 300 us at d_model 1024)
    const int rows = it.rg * rt, cols = cg * ct;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int g = row / rt, r = row - g * rt;
        const float* srow = src + ((size_t)g * rp + r) * cg * cp;
        float* drow = dst + (size_t)row * cols;
        for (int col = threadIdx.x; col < cols; col += blockDim.x) {
            const int c = (int)((unsigned)col / (unsigned)ct), kk = col - c * ct;
            const size_t j = (size_t)c * cp + kk;
            float v = srow[j];
            for (int z = 1; z < nslabs; ++z) v += srow[(size_t)z * stride + j];
            v *= alpha;
            if (accumulate) v += drow[col];
            drow[col] = v;
        }
    }
}

extern "C" int commu_reduce_slabs_group_f32(const commu_reduce_item* items, int nitems, const float* slabs, int nslabs,
                                            size_t stride, int accumulate, float alpha, hipStream_t stream) {
    if (nitems <= 0) return 0;
    if (nitems > 8 || nslabs <= 0) return -22;
    ReduceGroup grp;
    size_t nmax = 0;
    for (int i = 0; i < nitems; ++i) {
        grp.it[i] = items[i];
        if (items[i].rt > items[i].rp || items[i].ct > items[i].cp || items[i].rg <= 0 || items[i].cg <= 0) return -22;
        const size_t n = (size_t)items[i].rg * items[i].rt * items[i].cg * items[i].ct;
        nmax = n > nmax ? n : nmax;
    }
    size_t blocks = (nmax + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks == 0) return 0;
    COMMU_LAUNCH(reduce_slabs_group_kernel, dim3((unsigned)blocks, nitems), dim3(256), 0, stream, grp, slabs, nslabs, stride,
                 accumulate, alpha);
    COMMU_LAUNCH_CHECK();
    return 0;
}

// ---- grouped weight-gradient GEMM (gemm8.hip: gemm_tn8_kernel)
static bool tn_group_plan(const commu_tn_problem* probs, int nprob, int M, Tn8Args* out) {
    if (nprob <= 0 || nprob > 8) return false;
    int tiles = 0;
    for (int i = 0; i < nprob; ++i) {
        const commu_tn_problem& q = probs[i];
        if (!gemm8_tn_eligible(M, q.N, q.K, q.lda, q.ldb)) return false;
        Tn8Prob& t = out->p[i];
        t.A = (const bf16*)q.A; t.B = (const bf16*)q.B; t.lda = q.lda; t.ldb = q.ldb; t.N = q.N; t.Kc = q.K;
        t.tiles_n = (q.N + 255) / 256; t.tiles_k = (q.K + 255) / 256; t.tile0 = tiles; t.out_off = q.out_off;
        t.colsum_off = q.colsum_off;
        tiles += t.tiles_n * t.tiles_k;
    }
    out->nprob = nprob; out->M = M; out->total_tiles = tiles;
    return true;
}

extern "C" int commu_gemm_tn_grouped_slices_budget(const commu_tn_problem* probs, int nprob, int M, int budget) {
    Tn8Args a;
    if (budget <= 0 || !tn_group_plan(probs, nprob, M, &a)) return 0;
    int s = budget / a.total_tiles;
    if (s >= 8) s &= ~7;
    if (s < 1) s = 1;
    const int cap = (M + 1023) / 1024;          // at least 16 K-tiles per workgroup
    return s > cap ? cap : s;
}

extern "C" int commu_gemm_tn_grouped_slices(const commu_tn_problem* probs, int nprob, int M) {
    Tn8Args a;
    if (!tn_group_plan(probs, nprob, M, &a)) return 0;
    // tiles x slices <= 128 workgroups: HALF the CUs.  A workgroup of this kernel owns its CU's register file for its
    // whole life, and the launch runs on the side stream beside the backward pass: with 256 workgroups every main-stream
    // kernel issued meanwhile waits for CUs; with 128 (16 per XCD) half of every XCD stays free, the launch takes
    // twice as long and still ends well inside its layer (measured per step: 256 -> 16.55 ms, 128 -> 16.18, 192 -> 16.24,
    // 64 -> 16.63, 512 -> 16.73; reference default config 34.1 -> 32.6 ms).  Half as many slabs to reduce as well.
    int budget = 128;
    if (const char* e = getenv("COMMU_TN8_WGS")) budget = atoi(e);
    int s = budget / a.total_tiles;
    if (s >= 8) s &= ~7;
    if (s < 1) s = 1;
    const int cap = (M + 1023) / 1024;          // at least 16 K-tiles per workgroup
    if (s > cap) s = cap;
    return s;
}

extern "C" int commu_gemm_tn_bf16_grouped(const commu_tn_problem* probs, int nprob, int M, float* slabs,
                                          long long slab_stride, int nslices, hipStream_t stream) {
    Tn8Args a;
    if (nslices <= 0 || !tn_group_plan(probs, nprob, M, &a)) return -22;
    a.nslices = nslices;
    a.m_per_slice = (((M + nslices - 1) / nslices) + 63) / 64 * 64;
    a.slabs = slabs;
    a.slab_stride = slab_stride;
    return launch_gemm8_tn(a, stream);
}

/* decode-step Linear with the preceding LayerNorm fused in: C = LN(z)[M, K] . B[N, K]^T (+ bias, relu, resid);
 * a_out (optional) receives LN(z) as bf16 */
extern "C" int commu_gemm_nt_ln_bf16(const void* z, int ldz, const float* gamma, const float* beta, int D, float eps,
                                     void* a_out, int lda_out, const void* B, int ldb, void* C, int ldc, int M, int N,
                                     int K, const float* bias, const void* resid, int ldr, int flags,
                                     hipStream_t stream) {
    if (D <= 0 || D > K || (a_out != nullptr && (lda_out % 8))) return -22;
    const SkinnyLN ln{gamma, beta, (bf16*)a_out, lda_out, D, eps};
    return launch_gemm_nt(z, ldz, B, ldb, C, ldc, M, N, K, bias, resid, ldr, nullptr, 0, flags, 0u, 0.f, 1.f, 1,
                          GemmBatch{0, 0, 0, 0, 0, 0}, stream, &ln);
}

COMMU_DEFINE_SEED_SALT_SETTER(commu_seed_salt_gemm)
