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
#include "gemm8.h"
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

// the same with 4 bytes per lane: global (srd base + voff[lane]) -> LDS (lds_dst + 4 * lane)
__device__ __forceinline__ void dma4(srd_t srd, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(srd) : "memory");
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
        v = drop_keep(salted(a.drop_seed), (unsigned)m * (unsigned)a.N + (unsigned)n, a.drop_thr) ? v * a.drop_scale : 0.f;
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
        // Order: the two 64-byte halves of a row's 128-byte line leave in CONSECUTIVE store instructions (a line whose halves
        // arrive far apart is written back twice: 1.4x the output bytes at the memory interface, tests/probes/write_amp.sh).
        float bv[2][8];
        if (flags & COMMU_EPI_BIAS) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = nbase + 32 * j + 8 * g;
                const f32x4 b0 = *(const f32x4*)(a.bias + n), b1 = *(const f32x4*)(a.bias + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bv[j][e] = b0[e]; bv[j][4 + e] = b1[e]; }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // all of this row half's residual (or ReLU-mask: never both, see gemm8_nt_eligible) vectors are requested
            // before the first is used
            bf16x8 aux[8];
            if (flags & (COMMU_EPI_RESID | COMMU_EPI_RELUMASK)) {
                const bf16* ap = (flags & COMMU_EPI_RESID) ? a.resid : a.rmask;
                const int lda_ = (flags & COMMU_EPI_RESID) ? a.ldr : a.ldm;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    aux[q] = ld_bf16x8(ap + (size_t)(mbase + 16 * (4 * h + (q >> 1)) + r16) * lda_ + nbase + 32 * (q & 1) + 8 * g);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int mi = 4 * h + (q >> 1), j = q & 1;
                const int m = mbase + 16 * mi + r16, n = nbase + 32 * j + 8 * g;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[2 * j + (e >> 2)][mi][e & 3];
                if (flags & COMMU_EPI_BIAS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bv[j][e];
                }
                if (flags & COMMU_EPI_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (flags & COMMU_EPI_DROPOUT) {          // (N even, n a multiple of 8: one word per two columns)
                    const DropKey dk = drop_key(salted(a.drop_seed));
                    const unsigned q0 = ((unsigned)m * (unsigned)N + (unsigned)n) >> 1, thr_hi = a.drop_thr << 16;
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const unsigned w = drop_word(q0 + (unsigned)e2, dk);
                        v[2 * e2] = (unsigned short)w >= (unsigned short)a.drop_thr ? v[2 * e2] * a.drop_scale : 0.f;
                        v[2 * e2 + 1] = w >= thr_hi ? v[2 * e2 + 1] * a.drop_scale : 0.f;
                    }
                }
                if (flags & COMMU_EPI_RESID) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf2f(aux[q][e]);
                }
                if (flags & COMMU_EPI_RELUMASK) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf2f(aux[q][e]) > 0.f) ? v[e] * a.mask_scale : 0.f;
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

// Pipelined epilogue: an interior wave's outputs leave through bias -> relu -> dropout as 16-byte pieces per lane, one ROW HALF
// (64 rows x both 32-column halves: two quadrants) in phase 1 and the other in phase 3 of the NEXT tile's first K-tile --
// phase p's MFMAs are the first to touch quadrant p again, so quadrants 0 and 1 are still intact in phase 1 and 2, 3 in phase 3 --
// with the accumulators cleared for the tile that is already being accumulated: the conversion, the hash and the store issue
// of one wave row run under the other row's MFMAs instead of in a burst with the matrix pipe idle.
__device__ __forceinline__ bool g8_interior(const G8Args& a, int mbase, int nbase) {
    return (mbase + 128 <= a.M) && (nbase + 64 <= a.N) && (a.ldc % 8) == 0;
}

// (one 16-row block m4 of the quadrant)
template <bool OUT_F32, int J, int XH, int BITS>
__device__ __forceinline__ void g8_drain_piece(const G8Args& a, int flags, f32x4 (&acc)[4][8], const float (&bv)[8], int mbase,
                                               int nbase, int r16, int g, int m4, unsigned& word, unsigned bits_in) {
    const int N = a.N;
    const int n = nbase + 32 * J + 8 * g;
    {
        const int mi = 4 * XH + m4;
        const int m = mbase + 16 * mi + r16;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[2 * J + (e >> 2)][mi][e & 3];
        if (flags & COMMU_EPI_BIAS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[e];
        }
        if (flags & COMMU_EPI_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (flags & COMMU_EPI_DROPOUT) {
            // one hash word per TWO elements (common.h drop_word): the lane's 8 columns start at an even flat index (n is a
            // multiple of 8 and the launcher only takes an even N with COMMU_EPI_DROPOUT): four words for eight elements
            const DropKey dk = drop_key(salted(a.drop_seed));
            const unsigned q0 = ((unsigned)m * (unsigned)N + (unsigned)n) >> 1;
            const unsigned thr_hi = a.drop_thr << 16;          // (high half: whole-word compare; low half: 16-bit compare)
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const unsigned w = drop_word(q0 + (unsigned)e2, dk);
                v[2 * e2] = (unsigned short)w >= (unsigned short)a.drop_thr ? v[2 * e2] * a.drop_scale : 0.f;
                v[2 * e2 + 1] = w >= thr_hi ? v[2 * e2 + 1] * a.drop_scale : 0.f;
            }
        }
        if (BITS == 2) {          // ReLU backward: bit ? C * mask_scale : 0  (sign-extended bit AND-ed onto the product)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int keep = __builtin_amdgcn_sbfe((int)bits_in, 8 * m4 + e, 1);
                v[e] = __builtin_bit_cast(float, __builtin_bit_cast(int, v[e] * a.mask_scale) & keep);
            }
        }
        if (OUT_F32) {
            float* C = (float*)a.C + (size_t)m * a.ldc + n;
            *(f32x4*)C = (f32x4){v[0], v[1], v[2], v[3]};
            *(f32x4*)(C + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        } else {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
            st_bf16x8((bf16*)a.C + (size_t)m * a.ldc + n, o);
            if (BITS == 1) {          // what the NEXT reader of C sees as > 0: the rounded value
#pragma unroll
                for (int e = 0; e < 8; ++e) word |= (bf2f(o[e]) > 0.f ? 1u : 0u) << (8 * m4 + e);
            }
        }
        acc[2 * J][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[2 * J + 1][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// Both column halves of row half XH, row block by row block: the two 64-byte halves of an output row's 128-byte line leave in
// consecutive store instructions (halves that arrive a phase apart are written back twice for 11-20 % of the lines,
// tests/probes/write_amp.sh).  Quadrant numbers (phase order of the MFMAs): (J0,XH0) 0, (J1,XH0) 1, (J1,XH1) 2, (J0,XH1) 3.
template <bool OUT_F32, int XH, int BITS = 0>
__device__ __forceinline__ void g8_drain_rows(const G8Args& a, int flags, f32x4 (&acc)[4][8], const float (&bv)[2][8], int mbase,
                                              int nbase, int r16, int g, unsigned* bits_out, unsigned bits_in0, unsigned bits_in1) {
    unsigned word0 = 0u, word1 = 0u;
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4) {
        g8_drain_piece<OUT_F32, 0, XH, BITS>(a, flags, acc, bv[0], mbase, nbase, r16, g, m4, word0, bits_in0);
        g8_drain_piece<OUT_F32, 1, XH, BITS>(a, flags, acc, bv[1], mbase, nbase, r16, g, m4, word1, bits_in1);
    }
    if (BITS == 1) {
        bits_out[64 * (XH ? 3 : 0)] = word0;
        bits_out[64 * (XH ? 2 : 1)] = word1;
    }
}

// PIPE: the finished tile of an interior wave is written during the next tile's first K-tile (g8_drain_rows; epilogues without an
// auxiliary operand) instead of in one burst
// (PIPE 2: + one bit per output, (C > 0), to the word buffer a.rmask; PIPE 3: ReLU backward from such a buffer -- the
//  launcher guarantees that every wave is interior.  Word of (tile, wave, quadrant q in phase order, lane):
//  ((tile * 8 + wave) * 4 + q) * 64 + lane.)
template <bool OUT_F32, int ABL = 0, int PIPE = 0>
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
    const srd_t srdA = mk_srd(a.A, 0, a.M, a.lda, a.K), srdB = mk_srd(a.B, 0, a.N, a.ldb, a.K);
    auto stage = [&](int q, const Cur& c, int pb) {
        if (ABL == 2 && c.i + c.kt > 1) return;
        const unsigned dst = lds0 + pb * BUF_BYTES + q * HT_BYTES + w * 2048;
        // (ONE descriptor per operand for the whole kernel; the tile's first row travels in the scalar offset with the K offset:
        //  a descriptor per call cost ~20 scalar instructions and two branches, four times per K-tile)
        if (q == Q_XLO || q == Q_XHI) {
            const unsigned soff = (unsigned)(c.m0 + (q == Q_XHI ? 64 : 0)) * (unsigned)a.lda * 2u + (unsigned)c.kt * 128u;
            dma16(srdA, voffX[0], soff, dst);
            dma16(srdA, voffX[1], soff, dst + 1024);
        } else {
            const unsigned soff = (unsigned)(c.n0 + (q == Q_WHI ? 32 : 0)) * (unsigned)a.ldb * 2u + (unsigned)c.kt * 128u;
            dma16(srdB, voffW[0], soff, dst);
            dma16(srdB, voffW[1], soff, dst + 1024);
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
    // pipelined epilogue (PIPE): the finished tile of an interior wave is drained during the next tile's first K-tile
    bool pend = false;
    int pmb = 0, pnb = 0, bnb = -1;
    const int pflags = a.flags;
    constexpr int BITS = PIPE == 2 ? 1 : (PIPE == 3 ? 2 : 0);
    unsigned* pbits = nullptr;          // PIPE 2 / 3: this lane's word of quadrant 0 of the pending tile (quadrant q: + 64 q)
    // PIPE 3: the pending tile's four words per lane come to this wave's (otherwise unused) edge scratch by LDS-DMA at the end
    // of the tile, [q][lane] -- not to registers: the compiler may copy an asm-loaded register before the data has landed
    const LDS_AS unsigned* rbl = (const LDS_AS unsigned*)(smem + 2 * BUF_BYTES + w * 4096) + lane;
    const srd_t srdBits = __builtin_amdgcn_make_buffer_rsrc((void*)a.rmask, 0, PIPE == 3 ? (int)((size_t)a.M * a.N / 8) : 0, 0x00020000);
    float bvl[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) bvl[j][e] = 0.f;
    auto load_bias = [&](int nb) {          // this wave's 2 x 8 bias values per lane (reloaded only when the tile column changes)
        if (!(a.flags & COMMU_EPI_BIAS) || nb == bnb || nb + 64 > a.N) return;
        bnb = nb;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 b0 = *(const f32x4*)(a.bias + nb + 32 * j + 8 * g), b1 = *(const f32x4*)(a.bias + nb + 32 * j + 8 * g + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bvl[j][e] = b0[e]; bvl[j][4 + e] = b1[e]; }
        }
        // consumed HERE as far as the compiler can tell: its s_waitcnt for these loads lands in this (rare) branch and not,
        // as vmcnt(0), in front of every drain of the main loop -- where it would empty the staging queue
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bvl[j][e]));
    };
    if (PIPE) load_bias(c0.n0 + wc * 64);
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
        if (PIPE == 3 && pend) {          // the pending tile's sign words: older than this phase's Wlo pieces only
            if (v1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (PIPE && pend) g8_drain_rows<OUT_F32, 0, BITS>(a, pflags, acc, bvl, pmb, pnb, r16, g, pbits, PIPE == 3 ? rbl[0] : 0u, PIPE == 3 ? rbl[64] : 0u);
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
        if (PIPE && pend) g8_drain_rows<OUT_F32, 1, BITS>(a, pflags, acc, bvl, pmb, pnb, r16, g, pbits, PIPE == 3 ? rbl[192] : 0u, PIPE == 3 ? rbl[128] : 0u);
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
        if (v2) stage(Q_XHI, c2, pb);
        if (PIPE && pend) {
            // (vmcnt retires in order: the K-tile needed next was complete with phase 1's Wlo pieces; younger than those are the
            //  three half-tiles of phases 2-4 and the 4 x 4 (fp32 output: 4 x 8; with sign words: 4 x 5) output stores of the
            //  four drains)
            if (!v2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (OUT_F32) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
            else if (PIPE == 2) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            pend = false;
        } else if (v2) {
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
            const int mb = c0.m0 + wr * 128, nb = c0.n0 + wc * 64;
            if (PIPE && g8_interior(a, mb, nb) && !(a.flags & (COMMU_EPI_RESID | COMMU_EPI_RELUMASK))) {
                pend = true;
                pmb = mb;
                pnb = nb;
                if (PIPE) load_bias(nb);
                if (PIPE >= 2) {
                    const int lid = (c0.m0 >> 8) * a.tiles_n + (c0.n0 >> 8);
                    pbits = (unsigned*)a.rmask + ((size_t)(lid * 8 + w) * 4) * 64 + lane;
                    if (PIPE == 3) {
                        const unsigned vo = (unsigned)(((lid * 8 + w) * 4) * 64 + lane) * 4u;
                        const unsigned dst = lds0 + 2 * BUF_BYTES + w * 4096;
#pragma unroll
                        for (int q = 0; q < 4; ++q) dma4(srdBits, vo + 256u * q, dst + 256u * q);
                    }
                }
            } else {
                g8_store<OUT_F32>(a, acc, mb, nb, r16, g, (LDS_AS float*)(smem + 2 * BUF_BYTES + w * 4096), lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        c0 = c1;
        c1 = c2;
        advance(c2);
        pb ^= 1;
    }
    if (PIPE && pend) {          // the last tile of this workgroup
        if (PIPE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g8_drain_rows<OUT_F32, 0, BITS>(a, pflags, acc, bvl, pmb, pnb, r16, g, pbits, PIPE == 3 ? rbl[0] : 0u, PIPE == 3 ? rbl[64] : 0u);
        g8_drain_rows<OUT_F32, 1, BITS>(a, pflags, acc, bvl, pmb, pnb, r16, g, pbits, PIPE == 3 ? rbl[192] : 0u, PIPE == 3 ? rbl[128] : 0u);
    }
    if (wr == 0) G8_BAR();          // pairs with the stagger barrier of the second wave row
#undef G8_FRAG
#undef G8_MFMA
}

// ------------------------------------------------------------------------------------------------------------
// gemm_tn8: the same eight-phase pipeline for the weight gradients dW[n, k] = sum_m dY[m, n] * X[m, k]
// (autograd of every nn.Linear of commu/model/model.py:163-169,205,212,278; contraction over the tokens).
//  * Both operands have the contraction index m as their ROW index: a K-tile is 64 rows (m) of two 256-column
//    images; half-tiles are [64 m][128 columns] (256-byte rows: lo = tile columns 0..127, hi = 128..255), so every
//    DMA piece (4 rows x 256 bytes) reads whole 128-byte lines.  MFMA fragments (8 consecutive m of one column) come
//    from ds_read_b64_tr_b16 transpose reads; 32-byte unit index XOR (m & 3) | ((m >> 3) & 1) << 2 keeps the 8 rows of
//    a half-wave on 8 different bank groups.
//  * One launch covers up to 8 problems (all weight gradients of a layer) x nslices token slices; workgroup =
//    (slice, tile), slice-major ids + the XCD remap put every slice on ONE XCD: the 32 workgroups of an XCD
//    read the same token rows (of all problems) at the same time -> each activation byte leaves HBM once.
//  * Output: fp32 slab [slice][problem block] (summed by commu_reduce_slabs_f32), 16-byte stores.
struct TnTile {
    int prob, n0, k0;
};

__global__ __launch_bounds__(512) void gemm_tn8_kernel(const Tn8Args a) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3, r16 = lane & 15, g = lane >> 4;

    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int slice = lid / a.total_tiles, tile = lid - slice * a.total_tiles;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < a.nprob && tile >= a.p[i].tile0) pi = i;
    // (uniform select of the problem record; the fields are kernel arguments -> scalar registers)
    const bf16* Ap = a.p[0].A;
    const bf16* Bp = a.p[0].B;
    int lda = a.p[0].lda, ldb = a.p[0].ldb, N = a.p[0].N, Kc = a.p[0].Kc, tiles_k = a.p[0].tiles_k, tile0 = 0;
    long long out_off = a.p[0].out_off, colsum_off = a.p[0].colsum_off;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (pi == i) {
            Ap = a.p[i].A; Bp = a.p[i].B; lda = a.p[i].lda; ldb = a.p[i].ldb; N = a.p[i].N; Kc = a.p[i].Kc;
            tiles_k = a.p[i].tiles_k; tile0 = a.p[i].tile0; out_off = a.p[i].out_off; colsum_off = a.p[i].colsum_off;
        }
    const int tl = tile - tile0;
    const int n0 = (tl / tiles_k) * 256, k0 = (tl % tiles_k) * 256;
    const int mbeg = slice * a.m_per_slice;
    const int mend = min(a.M, mbeg + a.m_per_slice);
    const int nsteps = mend > mbeg ? (mend - mbeg + 63) >> 6 : 0;

    // ---- staging: piece j of wave w = half-tile rows 8 w + 4 j + (lane >> 4); LDS slot lane & 15 of the 256-byte row
    // holds logical 16-byte chunk c: unit (c >> 1) = (slot >> 1) ^ f(row)
    unsigned voffX[2][2], voffW[2][2];          // [lo / hi][piece]; columns beyond N / Kc are pushed out of range
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 8 * w + 4 * j + (lane >> 4);
        const int slot = lane & 15;
        const int f = (row & 3) | (((row >> 3) & 1) << 2);
        const int c = ((((slot >> 1) ^ f) << 1) | (slot & 1)) * 8;          // first column of the chunk inside the half
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            voffX[hi][j] = (n0 + 128 * hi + c < N) ? (unsigned)(row * lda + c) * 2u : 0x80000000u;
            voffW[hi][j] = (k0 + 128 * hi + c < Kc) ? (unsigned)(row * ldb + c) * 2u : 0x80000000u;
        }
    }
    const int rows_left = mend - mbeg;
    auto mk = [&](const bf16* base, int ld, int col0) {
        const unsigned nrec = rows_left > 0 ? (unsigned)(((size_t)(rows_left - 1) * ld + 128) * 2) : 0u;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)mbeg * ld + col0), 0, (int)nrec, 0x00020000);
    };
    const srd_t sXlo = mk(Ap, lda, n0), sXhi = mk(Ap, lda, n0 + 128), sWlo = mk(Bp, ldb, k0), sWhi = mk(Bp, ldb, k0 + 128);
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
    auto stage = [&](int q, int step, int pb) {
        const unsigned dst = lds0 + pb * BUF_BYTES + q * HT_BYTES + w * 2048;
        if (q == Q_XLO) { const unsigned so = (unsigned)step * 128u * (unsigned)lda; dma16(sXlo, voffX[0][0], so, dst); dma16(sXlo, voffX[0][1], so, dst + 1024); }
        else if (q == Q_XHI) { const unsigned so = (unsigned)step * 128u * (unsigned)lda; dma16(sXhi, voffX[1][0], so, dst); dma16(sXhi, voffX[1][1], so, dst + 1024); }
        else if (q == Q_WLO) { const unsigned so = (unsigned)step * 128u * (unsigned)ldb; dma16(sWlo, voffW[0][0], so, dst); dma16(sWlo, voffW[0][1], so, dst + 1024); }
        else { const unsigned so = (unsigned)step * 128u * (unsigned)ldb; dma16(sWhi, voffW[1][0], so, dst); dma16(sWhi, voffW[1][1], so, dst + 1024); }
    };

    // ---- fragments: lane (i = r16, g) supplies &img[32 ks + 8 g + (i >> 2) + 4 h][16-column block, + 4 (i & 3)]
    const LDS_AS char* lds = (const LDS_AS char*)smem;
    const int fi = ((r16 >> 2) & 3) | ((g & 1) << 2);
    const int lanebase = (8 * g + (r16 >> 2)) * 256 + (r16 & 3) * 8;
    int xoff[4], woff[2];
#pragma unroll
    for (int b = 0; b < 4; ++b) xoff[b] = lanebase + (((4 * wr + b) ^ fi) << 5);
#pragma unroll
    for (int b = 0; b < 2; ++b) woff[b] = lanebase + (((2 * wc + b) ^ fi) << 5);
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    auto frag = [&](int off) -> bf16x8 {          // two transpose reads: m = 8g .. 8g+3 and 8g+4 .. 8g+7
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(lds + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(lds + off + 1024));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // column sums of dY (a bias gradient) ride along: wave (wr, wc) multiplies its dY fragments of column block wc by a fragment
    // of ones -- 2 MFMAs beside the 32 of phases 1 and 3.  EVERY K-tile column of the problem does them and the first one writes:
    // the workgroups that share a dY column block must keep the same pace -- with the extra MFMAs on half of them the others ran
    // ahead, out of the L2's reach, and the laggards fetched their operands again (+13 % / +19 % of the launch's bytes with one /
    // two such problems in the group, tests/probes/tn8_reads.py)
    const bool csum = colsum_off >= 0;          // (every K-tile column does the work -- equal pace, see below -- the first one writes)
    f32x4 accb[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    typedef __attribute__((ext_vector_type(4))) unsigned tn_u32x4;
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (tn_u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
#define TN8_COLSUM(H)                                                                                                \
    if (csum) {                                                                                                      \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                             \
            if (wc == mi) {                                                                                          \
                accb[H] = mfma16(ones, xf[mi][0], accb[H]);                                                          \
                accb[H] = mfma16(ones, xf[mi][1], accb[H]);                                                          \
            }                                                                                                        \
    }

    if (nsteps > 0) {
        stage(Q_XLO, 0, 0);
        stage(Q_WHI, 0, 0);
        stage(Q_XHI, 0, 0);
        stage(Q_WLO, 0, 0);
        if (nsteps > 1) {
            stage(Q_XLO, 1, 1);
            stage(Q_WHI, 1, 1);
            stage(Q_XHI, 1, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    G8_BAR();
    if (wr == 1) G8_BAR();

    bf16x8 xf[4][2], wf[2][2];
    for (int st = 0; st < nsteps; ++st) {
        const int pb = st & 1, pbo = pb * BUF_BYTES;
        const bool v1 = st + 1 < nsteps, v2 = st + 2 < nsteps;
        // ---- phase 1: (Xlo, Wlo)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            wf[b][0] = frag(pbo + Q_WLO * HT_BYTES + woff[b]);
            wf[b][1] = frag(pbo + Q_WLO * HT_BYTES + woff[b] + 8192);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            xf[b][0] = frag(pbo + Q_XLO * HT_BYTES + xoff[b]);
            xf[b][1] = frag(pbo + Q_XLO * HT_BYTES + xoff[b] + 8192);
        }
        if (v1) stage(Q_WLO, st + 1, pb ^ 1);
        G8_WAIT_LGKM();
        G8_BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16(wf[ni][kk], xf[mi][kk], acc[ni][mi]);
        TN8_COLSUM(0)
        __builtin_amdgcn_s_setprio(0);
        G8_BAR();
        // ---- phase 2: (Xlo, Whi)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            wf[b][0] = frag(pbo + Q_WHI * HT_BYTES + woff[b]);
            wf[b][1] = frag(pbo + Q_WHI * HT_BYTES + woff[b] + 8192);
        }
        if (v2) stage(Q_XLO, st + 2, pb);
        G8_WAIT_LGKM();
        G8_BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[2 + ni][mi] = mfma16(wf[ni][kk], xf[mi][kk], acc[2 + ni][mi]);
        __builtin_amdgcn_s_setprio(0);
        G8_BAR();
        // ---- phase 3: (Xhi, Whi)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            xf[b][0] = frag(pbo + Q_XHI * HT_BYTES + xoff[b]);
            xf[b][1] = frag(pbo + Q_XHI * HT_BYTES + xoff[b] + 8192);
        }
        if (v2) stage(Q_WHI, st + 2, pb);
        G8_WAIT_LGKM();
        G8_BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[2 + ni][4 + mi] = mfma16(wf[ni][kk], xf[mi][kk], acc[2 + ni][4 + mi]);
        TN8_COLSUM(1)
        __builtin_amdgcn_s_setprio(0);
        G8_BAR();
        // ---- phase 4: (Xhi, Wlo)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            wf[b][0] = frag(pbo + Q_WLO * HT_BYTES + woff[b]);
            wf[b][1] = frag(pbo + Q_WLO * HT_BYTES + woff[b] + 8192);
        }
        if (v2) {
            stage(Q_XHI, st + 2, pb);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
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
                for (int mi = 0; mi < 4; ++mi) acc[ni][4 + mi] = mfma16(wf[ni][kk], xf[mi][kk], acc[ni][4 + mi]);
        __builtin_amdgcn_s_setprio(0);
        G8_BAR();
    }
    if (wr == 0) G8_BAR();

    // ---- slab store: lane holds out[n = .. + r16][k = .. + 4 g + reg]
    float* out = a.slabs + (size_t)slice * a.slab_stride + out_off;
#pragma unroll
    for (int xb = 0; xb < 8; ++xb) {
        const int n = n0 + 128 * (xb >> 2) + 64 * wr + 16 * (xb & 3) + r16;
        if (n >= N) continue;
#pragma unroll
        for (int wb = 0; wb < 4; ++wb) {
            const int k = k0 + 128 * (wb >> 1) + 32 * wc + 16 * (wb & 1) + 4 * g;
            if (k < Kc) *(f32x4*)(out + (size_t)n * Kc + k) = acc[wb][xb];
        }
    }
    if (csum && k0 == 0 && g == 0) {          // every row of the ones product is the column sum: row 0 (lanes g == 0, register 0)
        float* cs = a.slabs + (size_t)slice * a.slab_stride + colsum_off;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int n = n0 + 128 * hh + 64 * wr + 16 * wc + r16;
            if (n < N) cs[n] = accb[hh][0];
        }
    }
#undef TN8_COLSUM
}

}  // namespace

bool gemm8_nt_eligible(int M, int N, int K, int lda, int ldb, int batch, int tri_B, int flags) {
    if (getenv("COMMU_GEMM8_OFF")) return false;
    if (batch != 1 || tri_B != 0) return false;
    if ((flags & COMMU_EPI_RESID) && (flags & COMMU_EPI_RELUMASK)) return false;          // one auxiliary operand per call
    if (M < 1024 || N < 256 || K < 128 || (K % 64) != 0) return false;
    if ((flags & COMMU_EPI_DROPOUT) && (N & 1)) return false;          // the pipelined epilogue pairs columns 2k, 2k+1 per hash word
    // fewer output tiles than CUs: the persistent kernel's fixed cost (~20 us) is not amortised and three quarters of the
    // chip idle -- the 128 x 128 / 256 x 128 tiled kernels take these (8192 x 512 x 512: 14 us against 22 us)
    if (((M + 255) / 256) * ((N + 255) / 256) < 256 && !getenv("COMMU_GEMM8_ALWAYS")) return false;
    if ((size_t)M * lda * 2 >= 0xFFFF0000ull || (size_t)N * ldb * 2 >= 0xFFFF0000ull) return false;      // 32-bit buffer offsets
    if ((size_t)256 * lda * 2 >= 0x7FFF0000ull || (size_t)256 * ldb * 2 >= 0x7FFF0000ull) return false;
    return true;
}

bool gemm8_nt_bits_eligible(int M, int N, int K, int lda, int ldb, int ldc, int flags) {
    // every wave of every tile takes the pipelined (interior) epilogue: whole tiles, 16-byte aligned output rows, bf16 output,
    // no auxiliary operand; ALWAYS in the environment (tests) lifts the 256-tile minimum of gemm8_nt_eligible
    if ((M % 256) || (N % 256) || (ldc % 8)) return false;
    if (flags & (COMMU_EPI_RESID | COMMU_EPI_RELUMASK | COMMU_EPI_OUT_F32)) return false;
    if ((flags & COMMU_EPI_SIGNBITS_OUT) && (flags & COMMU_EPI_RELUBITS)) return false;
    return gemm8_nt_eligible(M, N, K, lda, ldb, 1, 0, flags);
}

int launch_gemm8_nt(const G8Args& a_in, hipStream_t stream) {
    const G8Args& a0 = a_in;
    const int ntiles = a0.tiles_m * a0.tiles_n;
    int grid = ntiles < 256 ? ntiles : 256;
    if (const char* e = getenv("COMMU_GEMM8_GRID")) {
        const int gsz = atoi(e);
        if (gsz > 0 && gsz < grid) grid = gsz;
    }
    G8Args a = a_in;
    a.skew_cycles = 0;
    if (const char* e = getenv("COMMU_GEMM8_SKEW")) a.skew_cycles = atoi(e);
    a.store_mode = 0;
    if (const char* e = getenv("COMMU_GEMM8_ST")) a.store_mode = atoi(e);
#define G8_LAUNCH(F32, AB) COMMU_LAUNCH((gemm_nt8_kernel<F32, AB>), dim3(grid), dim3(512), 0, stream, a)
    const bool pipe = getenv("COMMU_GEMM8_NOPIPE") == nullptr;
    if (a.flags & COMMU_EPI_OUT_F32) {
        G8_LAUNCH(true, 0);
    } else if (a.flags & COMMU_EPI_SIGNBITS_OUT) {
        COMMU_LAUNCH((gemm_nt8_kernel<false, 0, 2>), dim3(grid), dim3(512), 0, stream, a);
    } else if (a.flags & COMMU_EPI_RELUBITS) {
        COMMU_LAUNCH((gemm_nt8_kernel<false, 0, 3>), dim3(grid), dim3(512), 0, stream, a);
    } else if (pipe && !(a.flags & (COMMU_EPI_RESID | COMMU_EPI_RELUMASK))) {
        COMMU_LAUNCH((gemm_nt8_kernel<false, 0, 1>), dim3(grid), dim3(512), 0, stream, a);
    } else {
        G8_LAUNCH(false, 0);          // (ABL != 0 -- fake MFMA / no staging / no epilogue -- are profiling builds: never launched)
    }
#undef G8_LAUNCH
    COMMU_LAUNCH_CHECK();
    return 0;
}

bool gemm8_tn_eligible(int M, int N, int Kc, int lda, int ldb) {
    if (getenv("COMMU_TN8_OFF")) return false;
    if (M < 4096 || N < 128 || Kc < 128 || (N % 8) || (Kc % 8) || (lda % 8) || (ldb % 8)) return false;
    if (lda < 128 || ldb < 128) return false;
    if ((size_t)M * lda * 2 >= 0x7FFF0000ull || (size_t)M * ldb * 2 >= 0x7FFF0000ull) return false;
    return true;
}

int launch_gemm8_tn(const Tn8Args& a, hipStream_t stream) {
    if (a.total_tiles <= 0 || a.nslices <= 0) return 0;
    COMMU_LAUNCH(gemm_tn8_kernel, dim3(a.total_tiles * a.nslices), dim3(512), 0, stream, a);
    COMMU_LAUNCH_CHECK();
    return 0;
}

COMMU_DEFINE_SEED_SALT_SETTER(commu_seed_salt_gemm8)
