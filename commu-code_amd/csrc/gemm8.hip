// 256 x 256 x 64 eight-phase bf16 GEMM for the large-M Linear shapes of the training step (gfx950).
//
//   gemm_nt8 : C[M,N] = A[M,K] . B[N,K]^T (+ the fused epilogue of gemm.hip)   -- nn.Linear forward and dX = dY . W
//              (reference: commu/model/model.py:205,212,164,167,46 and their autograd)
//
// Structure (one 512-thread workgroup per CU, persistent over output tiles):
//  * 8 waves = 2 (M) x 4 (N); a wave owns 128 x 64 outputs = four 64 x 32 QUADRANTS (X lo/hi rows x W lo/hi columns).
//  * A K-tile (64 deep) is FOUR 16-KB half-tiles in LDS: Xlo, Xhi (the lo / hi 64 rows of both wave rows),
//    Wlo, Whi (the lo / hi 32 columns of all four wave columns); two K-tile buffers = 128 KB.
//  * A K-tile is computed in 4 phases, one quadrant (16 MFMA 16x16x32) each, in the order (Xlo,Wlo) (Xlo,Whi)
//    (Xhi,Whi) (Xhi,Wlo): exactly one half-tile dies per phase (Xlo, Whi, Xhi, Wlo), and exactly one half-tile
//    of the K-tile two ahead is staged per phase into the slot that just died -> three half-tiles are always in
//    flight; the only wait is a COUNTED s_waitcnt vmcnt(6) once per K-tile (phase 4).
//  * Staging is LDS-DMA (buffer_load_dwordx4 ... lds) issued from inline asm, so the compiler's own waitcnt
//    bookkeeping never drains the queue; out-of-range rows (M / N tails) read as zero through the buffer SRD.
//  * The two wave rows run staggered by one barrier: while one half of the workgroup issues its 16 MFMAs the other
//    half reads fragments / issues DMA, so each SIMD's matrix pipe always has a wave in its MFMA segment.
//  * LDS image: 1-KB pieces of 8 rows x 128 bytes (one DMA instruction = 8 full 128-byte lines), 16-byte chunk
//    index XOR (row >> 1) & 7: every ds_read_b128 lane group touches 16 distinct 16-byte slots (conflict-free).
//  * The load stream runs continuously across output tiles: the next tile's first K-tiles are in flight while the
//    finished tile is written out.
#include "gemm8.cuh"
#include "commu_hip.h"
#include <stdlib.h>

namespace {

typedef __amdgpu_buffer_rsrc_t srd_t;

constexpr int HT_BYTES = 16384;              // half-tile: 128 rows x 64 k, bf16
constexpr int BUF_BYTES = 4 * HT_BYTES;      // one K-tile
constexpr int Q_XLO = 0, Q_XHI = 1, Q_WLO = 2, Q_WHI = 3;

// one LDS-DMA piece: 64 lanes x 16 bytes, global (srd base + voff[lane] + soff) -> LDS (lds_dst + 16 * lane)
__device__ __forceinline__ void dma16(srd_t srd, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// SRD over rows [row0, rows_total) of a row-major bf16 matrix with K columns: rows >= rows_total lie beyond
// num_records and read as zero (the K offset travels in soffset; gfx950 range-checks voffset + soffset).
__device__ __forceinline__ srd_t mk_srd(const bf16* base, int row0, int rows_total, int ld, int K) {
    const int left = rows_total - row0;
    const unsigned nrec = left > 0 ? (unsigned)(((size_t)(left - 1) * ld + K) * 2) : 0u;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)row0 * ld), 0, (int)nrec, 0x00020000);
}

// (ablation) keeps the operands alive without issuing an MFMA
__device__ __forceinline__ f32x4 g8_fake(bf16x8 wa, bf16x8 xb, f32x4 c) {
    asm volatile("" :: "v"(wa), "v"(xb));
    return c;
}

struct Cur {          // position of a K-tile in this workgroup's flattened (output tile, K-tile) stream
    int i, kt, m0, n0;
};

#define G8_BAR()                               \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
#define G8_WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// One output element through the fused epilogue (bias -> relu -> dropout -> resid -> relu-mask), then stored.
template <bool OUT_F32>
__device__ __forceinline__ void g8_store_one(const G8Args& a, float v, int m, int n) {
    const int flags = a.flags;
    if (flags & COMMU_EPI_BIAS) v += a.bias[n];
    if (flags & COMMU_EPI_RELU) v = fmaxf(v, 0.f);
    if (flags & COMMU_EPI_DROPOUT)
        v = drop_keep(a.drop_seed, (unsigned)m * (unsigned)a.N + (unsigned)n, a.drop_thr) ? v * a.drop_scale : 0.f;
    if (flags & COMMU_EPI_RESID) v += bf2f(a.resid[(size_t)m * a.ldr + n]);
    if (flags & COMMU_EPI_RELUMASK) v = (bf2f(a.rmask[(size_t)m * a.ldm + n]) > 0.f) ? v * a.mask_scale : 0.f;
    if (OUT_F32) ((float*)a.C)[(size_t)m * a.ldc + n] = v;
    else ((bf16*)a.C)[(size_t)m * a.ldc + n] = f2bf(v);
}

// Write a wave's 128 x 64 outputs.  Interior waves (no M / N tail, 16-byte aligned rows): every lane owns 8
// consecutive columns per (row block, column half) -> 16-byte loads / stores, no predicates.  Edge waves: the
// accumulators go through a wave-private 4-KB LDS scratch, 16 rows at a time, and are stored one element per lane
// (compact code; only the last row / column of tiles takes this path).
template <bool OUT_F32>
__device__ __forceinline__ void g8_store(const G8Args& a, f32x4 (&acc)[4][8], int mbase, int nbase, int r16, int g,
                                         LDS_AS float* scratch, int lane) {
    const int flags = a.flags, N = a.N;
    const bool interior = (mbase + 128 <= a.M) && (nbase + 64 <= N) && (a.ldc % 8) == 0 &&
                          (!(flags & COMMU_EPI_RESID) || (a.ldr % 8) == 0) &&
                          (!(flags & COMMU_EPI_RELUMASK) || (a.ldm % 8) == 0);
    if (interior) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nbase + 32 * j + 8 * g;
            float bv[8];
            if (flags & COMMU_EPI_BIAS) {
                const f32x4 b0 = *(const f32x4*)(a.bias + n), b1 = *(const f32x4*)(a.bias + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
            }
            // all of this column half's residual (or ReLU-mask: never both, see gemm8_nt_eligible) vectors are
            // requested before the first is used
            bf16x8 aux[8];
            if (flags & (COMMU_EPI_RESID | COMMU_EPI_RELUMASK)) {
                const bf16* ap = (flags & COMMU_EPI_RESID) ? a.resid : a.rmask;
                const int lda_ = (flags & COMMU_EPI_RESID) ? a.ldr : a.ldm;
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) aux[mi] = ld_bf16x8(ap + (size_t)(mbase + 16 * mi + r16) * lda_ + n);
            }
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int m = mbase + 16 * mi + r16;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[2 * j + (e >> 2)][mi][e & 3];
                if (flags & COMMU_EPI_BIAS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bv[e];
                }
                if (flags & COMMU_EPI_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (flags & COMMU_EPI_DROPOUT) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[e] = drop_keep(a.drop_seed, (unsigned)m * (unsigned)N + (unsigned)(n + e), a.drop_thr)
                                   ? v[e] * a.drop_scale : 0.f;
                }
                if (flags & COMMU_EPI_RESID) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf2f(aux[mi][e]);
                }
                if (flags & COMMU_EPI_RELUMASK) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf2f(aux[mi][e]) > 0.f) ? v[e] * a.mask_scale : 0.f;
                }
                if (OUT_F32) {
                    float* C = (float*)a.C + (size_t)m * a.ldc + n;
                    *(f32x4*)C = (f32x4){v[0], v[1], v[2], v[3]};
                    *(f32x4*)(C + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
                    bf16* cp = (bf16*)a.C + (size_t)m * a.ldc + n;
                    if (a.store_mode == 1) __builtin_nontemporal_store(o, (bf16x8*)cp);
                    else if (a.store_mode == 2)
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(cp), "v"(o) : "memory");
                    else if (a.store_mode == 3)
                        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(cp), "v"(o) : "memory");
                    else st_bf16x8(cp, o);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            *(LDS_AS f32x4*)(scratch + r16 * 64 + 32 * (ni >> 1) + 8 * g + 4 * (ni & 1)) = acc[ni][mi];
        const int n = nbase + lane;
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            const float v = scratch[it * 64 + lane];
            const int m = mbase + 16 * mi + it;
            if (m < a.M && n < N) g8_store_one<OUT_F32>(a, v, m, n);
        }
    }
}

// ABL (profiling only, COMMU_GEMM8_ABL): 1 = no MFMA, 2 = no staging after the prologue, 3 = no output stores
template <bool OUT_F32, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_nt8_kernel(const G8Args a) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * BUF_BYTES + 8 * 4096];          // + edge-tile scratch
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3, r16 = lane & 15, g = lane >> 4;
    const int nk = a.K >> 6;

    // ---- this workgroup's output tiles: XCD x owns a contiguous range of tile ids (tile columns fastest, so the
    // workgroups of an XCD share activation row blocks in that XCD's L2); its workgroups stride through the range
    const int ntiles = a.tiles_m * a.tiles_n;
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    // (a grid that is not a multiple of 8 -- tests, tiny problems -- strides through the tile ids directly)
    const bool xcdmap = (G & 7) == 0;
    const int xcd = bid & 7, idx = xcdmap ? bid >> 3 : bid;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int xbase = !xcdmap ? 0 : (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8);
    const int xcount = !xcdmap ? ntiles : q8 + (xcd < r8 ? 1 : 0);
    const int cpx = xcdmap ? G >> 3 : G;
    const int my_n = idx < xcount ? (xcount - idx + cpx - 1) / cpx : 0;
    if (my_n == 0) return;

    auto settile = [&](Cur& c) {
        if (c.i < my_n) {
            const int lid = xbase + idx + c.i * cpx;
            const int tm = lid / a.tiles_n;
            c.m0 = tm * 256;
            c.n0 = (lid - tm * a.tiles_n) * 256;
        }
    };
    auto advance = [&](Cur& c) {
        if (++c.kt == nk) {
            c.kt = 0;
            ++c.i;
            settile(c);
        }
    };

    // ---- staging addresses: piece j of wave w fills half-tile rows h = 16 w + 8 j + (lane >> 3); LDS slot lane & 7 of
    // that row holds logical 16-byte chunk (lane & 7) ^ ((h >> 1) & 7)
    unsigned voffX[2], voffW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int h = 16 * w + 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ ((h >> 1) & 7);
        const int xrow = (h >> 6) * 128 + (h & 63);                                             // (hi half: + 64)
        const int wcol = (h >> 5) * 64 + 8 * ((h & 15) >> 2) + 4 * ((h >> 4) & 1) + (h & 3);    // (hi half: + 32)
        voffX[j] = (unsigned)(xrow * a.lda + c * 8) * 2u;
        voffW[j] = (unsigned)(wcol * a.ldb + c * 8) * 2u;
    }
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
    auto stage = [&](int q, const Cur& c, int pb) {
        if (ABL == 2 && c.i + c.kt > 1) return;
        const unsigned dst = lds0 + pb * BUF_BYTES + q * HT_BYTES + w * 2048;
        const unsigned soff = (unsigned)c.kt * 128u;
        if (q == Q_XLO || q == Q_XHI) {
            const srd_t s = mk_srd(a.A, c.m0 + (q == Q_XHI ? 64 : 0), a.M, a.lda, a.K);
            dma16(s, voffX[0], soff, dst);
            dma16(s, voffX[1], soff, dst + 1024);
        } else {
            const srd_t s = mk_srd(a.B, c.n0 + (q == Q_WHI ? 32 : 0), a.N, a.ldb, a.K);
            dma16(s, voffW[0], soff, dst);
            dma16(s, voffW[1], soff, dst + 1024);
        }
    };

    // ---- fragment addresses: block row R0 (multiple of 16) + r16, k-step kk: chunk (4 kk + g) ^ ((r16 >> 1) & 7)
    const LDS_AS char* lds = (const LDS_AS char*)smem;
    const int laneoff = (r16 >> 3) * 1024 + (r16 & 7) * 128;
    const int ch0 = (g ^ ((r16 >> 1) & 7)) << 4;
    const int xo0 = laneoff + ch0 + wr * 8192, xo1 = laneoff + (ch0 ^ 64) + wr * 8192;
    const int wo0 = laneoff + ch0 + wc * 4096, wo1 = laneoff + (ch0 ^ 64) + wc * 4096;
#define G8_FRAG(off) (*(const LDS_AS bf16x8*)(lds + (off)))
#define G8_MFMA(wa, xb, c) (ABL == 1 ? g8_fake(wa, xb, c) : mfma16(wa, xb, c))

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Cur c0{0, 0, 0, 0}, c1, c2;
    settile(c0);
    c1 = c0;
    advance(c1);
    c2 = c1;
    advance(c2);

    // ---- prologue: K-tile 0 entirely, three half-tiles of K-tile 1
    stage(Q_XLO, c0, 0);
    stage(Q_WHI, c0, 0);
    stage(Q_XHI, c0, 0);
    stage(Q_WLO, c0, 0);
    if (c1.i < my_n) {
        stage(Q_XLO, c1, 1);
        stage(Q_WHI, c1, 1);
        stage(Q_XHI, c1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (a.skew_cycles > 0) {
        // de-synchronise the CUs of an XCD (8 phase groups): without it every workgroup reaches its output stores at
        // the same moment and the kernel alternates between an idle and a saturated HBM write path
        const long long t0 = (long long)__builtin_readcyclecounter();
        const long long wait = (long long)((bid >> 3) & 7) * a.skew_cycles;
        while ((long long)__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    G8_BAR();
    if (wr == 1) G8_BAR();          // the second wave row runs one barrier behind the first

    bf16x8 xf[4][2], wf[2][2];
    int pb = 0;
    while (c0.i < my_n) {
        const int pbo = pb * BUF_BYTES;
        const bool v1 = c1.i < my_n, v2 = c2.i < my_n;
        // ---------------- phase 1: (Xlo, Wlo); stage Wlo of the NEXT K-tile
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            wf[ni][0] = G8_FRAG(pbo + Q_WLO * HT_BYTES + wo0 + ni * 2048);
            wf[ni][1] = G8_FRAG(pbo + Q_WLO * HT_BYTES + wo1 + ni * 2048);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            xf[mi][0] = G8_FRAG(pbo + Q_XLO * HT_BYTES + xo0 + mi * 2048);
            xf[mi][1] = G8_FRAG(pbo + Q_XLO * HT_BYTES + xo1 + mi * 2048);
        }
        if (v1) stage(Q_WLO, c1, pb ^ 1);
        G8_WAIT_LGKM();
        G8_BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = G8_MFMA(wf[ni][kk], xf[mi][kk], acc[ni][mi]);
        __builtin_amdgcn_s_setprio(0);
        G8_BAR();
        // ---------------- phase 2: (Xlo, Whi); Xlo of this buffer is dead -> stage Xlo two K-tiles ahead
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            wf[ni][0] = G8_FRAG(pbo + Q_WHI * HT_BYTES + wo0 + ni * 2048);
            wf[ni][1] = G8_FRAG(pbo + Q_WHI * HT_BYTES + wo1 + ni * 2048);
        }
        if (v2) stage(Q_XLO, c2, pb);
        G8_WAIT_LGKM();
        G8_BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[2 + ni][mi] = G8_MFMA(wf[ni][kk], xf[mi][kk], acc[2 + ni][mi]);
        __builtin_amdgcn_s_setprio(0);
        G8_BAR();
        // ---------------- phase 3: (Xhi, Whi); Whi is dead -> stage Whi two ahead
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            xf[mi][0] = G8_FRAG(pbo + Q_XHI * HT_BYTES + xo0 + mi * 2048);
            xf[mi][1] = G8_FRAG(pbo + Q_XHI * HT_BYTES + xo1 + mi * 2048);
        }
        if (v2) stage(Q_WHI, c2, pb);
        G8_WAIT_LGKM();
        G8_BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[2 + ni][4 + mi] = G8_MFMA(wf[ni][kk], xf[mi][kk], acc[2 + ni][4 + mi]);
        __builtin_amdgcn_s_setprio(0);
        G8_BAR();
        // ---------------- phase 4: (Xhi, Wlo); Xhi is dead -> stage Xhi two ahead; the next K-tile must have landed
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            wf[ni][0] = G8_FRAG(pbo + Q_WLO * HT_BYTES + wo0 + ni * 2048);
            wf[ni][1] = G8_FRAG(pbo + Q_WLO * HT_BYTES + wo1 + ni * 2048);
        }
        if (v2) {
            stage(Q_XHI, c2, pb);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // all but the three half-tiles staged in phases 2-4
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        G8_WAIT_LGKM();
        G8_BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[ni][4 + mi] = G8_MFMA(wf[ni][kk], xf[mi][kk], acc[ni][4 + mi]);
        __builtin_amdgcn_s_setprio(0);
        G8_BAR();
        // ---------------- end of an output tile: write it out (the next tile's loads are already in flight)
        if (c0.kt == nk - 1 && (ABL != 3 || acc[0][0][0] == 1234.5f)) {
            g8_store<OUT_F32>(a, acc, c0.m0 + wr * 128, c0.n0 + wc * 64, r16, g,
                             (LDS_AS float*)(smem + 2 * BUF_BYTES + w * 4096), lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        c0 = c1;
        c1 = c2;
        advance(c2);
        pb ^= 1;
    }
    if (wr == 0) G8_BAR();          // pairs with the stagger barrier of the second wave row
#undef G8_FRAG
#undef G8_MFMA
}

}  // namespace

bool gemm8_nt_eligible(int M, int N, int K, int lda, int ldb, int batch, int tri_B, int flags) {
    if (getenv("COMMU_GEMM8_OFF")) return false;
    if (batch != 1 || tri_B != 0) return false;
    if ((flags & COMMU_EPI_RESID) && (flags & COMMU_EPI_RELUMASK)) return false;          // one auxiliary operand per call
    if (M < 1024 || N < 256 || K < 128 || (K % 64) != 0) return false;
    if ((size_t)M * lda * 2 >= 0xFFFF0000ull || (size_t)N * ldb * 2 >= 0xFFFF0000ull) return false;      // 32-bit buffer offsets
    if ((size_t)256 * lda * 2 >= 0x7FFF0000ull || (size_t)256 * ldb * 2 >= 0x7FFF0000ull) return false;
    return true;
}

int launch_gemm8_nt(const G8Args& a_in, hipStream_t stream) {
    const G8Args& a0 = a_in;
    const int ntiles = a0.tiles_m * a0.tiles_n;
    int grid = ntiles < 256 ? ntiles : 256;
    if (const char* e = getenv("COMMU_GEMM8_GRID")) {
        const int gsz = atoi(e);
        if (gsz > 0 && gsz < grid) grid = gsz;
    }
    int abl = 0;
    if (const char* e = getenv("COMMU_GEMM8_ABL")) abl = atoi(e);
    G8Args a = a_in;
    a.skew_cycles = 0;
    if (const char* e = getenv("COMMU_GEMM8_SKEW")) a.skew_cycles = atoi(e);
    a.store_mode = 0;
    if (const char* e = getenv("COMMU_GEMM8_ST")) a.store_mode = atoi(e);
#define G8_LAUNCH(F32, AB) COMMU_LAUNCH((gemm_nt8_kernel<F32, AB>), dim3(grid), dim3(512), 0, stream, a)
    if (a.flags & COMMU_EPI_OUT_F32) {
        G8_LAUNCH(true, 0);
    } else {
        switch (abl) {
            case 1: G8_LAUNCH(false, 1); break;
            case 2: G8_LAUNCH(false, 2); break;
            case 3: G8_LAUNCH(false, 3); break;
            default: G8_LAUNCH(false, 0); break;
        }
    }
#undef G8_LAUNCH
    COMMU_LAUNCH_CHECK();
    return 0;
}
