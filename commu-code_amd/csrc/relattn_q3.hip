// K6 backward, query-stationary, on the 32x32x16 MFMA with the score tile TRANSPOSED (gfx950, d_head 64): the layout of
// relattn3.hip (forward) applied to the first backward kernel.  Reference: autograd of commu/model/model.py:313-345
// (_rel_shift :251-259, masks :549-574); same contract as relattn_bwd_q_kernel (relattn.hip):
//   dq_AC = dS . K  (+ its column sums per 64-row tile for d r_w_bias), dS written BY DISTANCE (dSk[i][d = i+M-j]) for the band
//   pass, and P scale/(1-p) (sign = dropped) for the key-stationary kernel.
//
// A wave owns 32 query rows and walks the keys in sub-tiles of 32.  With S^T = K . (q+u)^T the accumulator layout is
// lane = QUERY, 16 registers = KEYS jj = 8 (r >> 2) + 4 half + 3 - (r & 3): lse and delta are per-lane scalars (they are the
// INITIAL values of the band / dP accumulators), exp2, the dropout select and dS = P (dP - delta) are lane-local, and dS^T --
// converted to bf16 in place -- IS the B operand of dq^T += K^T . dS^T (K^T by transpose reads in the same key order).  The
// 16x16 kernel needed a per-wave dS^T image in LDS, sixteen ds_bpermute for the rel-shift and 620 instructions per 1024
// scores; this one issues about 270.
//   * rel-shift: the band product (q+v) . Rd^T comes out by DISTANCE; it reaches the key-indexed accumulators through the
//     per-wave fp32 ring of relattn3.hip (written at immediate offsets, read back as the initial value of the score MFMA).
//   * dS by distance: the inverse skew, through a per-wave bf16 ring [32 rows][64 distances] (2-byte writes at
//     (distance & 63), the and-or addressing of the 16x16 kernel), flushed as whole aligned 16-byte chunks, 4 per row and
//     sub-tile.
//   * P leaves in accumulator order (p_layout 2: 2-KB blocks of [32 queries x 32 keys], piece k of lane l at 1024 k + 16 l --
//     every store instruction writes one contiguous KB); relattn_bwd_kv3_kernel<.., true> brings a block to LDS by LDS-DMA
//     and transposes it with ds_read_b64_tr_b16.
// Workgroup = 4 waves = 128 query rows of one (batch, head), two workgroups per CU (69 KB of LDS each).  Only K goes through
// LDS (double-buffered 64-key tiles by LDS-DMA, one barrier per tile: it is read row-wise for S^T and transposed for dq^T);
// the V rows (A operand of dP^T = V . dO^T) and the Rd rows (A operand of the band product) are MFMA fragments loaded
// straight from global memory / L2 one sub-tile ahead -- a row is 128 contiguous bytes, every wave of the workgroup reads
// the same V rows.
#include "relattn_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int TILEB = 8192;                                   // [64 rows][64] bf16
constexpr int OFF_K = 0;                                      // K x2
constexpr int RINGB = 8832;                                   // per wave: BD ring, rows of 64 fp32 at word 68 i + 4 (i >> 2)
constexpr int OFF_RING = 2 * TILEB;
constexpr int DSB = 4096;                                     // per wave: dS-by-distance ring [32 rows][64 distances] bf16
constexpr int OFF_DS = OFF_RING + 4 * RINGB;                  // 51712 (a multiple of 128: the and-or addressing needs it)
constexpr int OFF_RED = OFF_DS + 4 * DSB;                     // column sums of dq per wave: [4][64] fp32
constexpr int LDS_Q3 = OFF_RED + 4 * 64 * 4;                  // 69120 bytes
static_assert(OFF_DS % 128 == 0, "dS ring rows start at multiples of 128 bytes");

// 16-byte chunk c of row R of a [64][64] bf16 tile lives at chunk c ^ swz3(R) (relattn3.hip)
__device__ __forceinline__ int swz3(int R) {
    const int p = R >> 1;
    return ((p & 1) << 2) | (p & 2) | ((p >> 2) & 1);
}

// the attention-dropout mask of relattn.hip (DropLane) in the transposed layout: see relattn3.hip
constexpr unsigned DROP_C1 = 0xD2B74Bu;
constexpr unsigned DROP_CM[2][2] = {{0x9E3779u, 0x85EBCBu}, {0xC2B2AFu, 0xB5297Bu}};
constexpr unsigned DROP_KA[2][2] = {{0x85EBCA6Bu, 0xC2B2AE35u}, {0x27D4EB2Fu, 0x165667B1u}};
constexpr unsigned DROP_KB[2][2] = {{0x6A09E667u, 0xBB67AE85u}, {0x3C6EF372u, 0xA54FF53Au}};

// ABL: profiling ablations (COMMU_Q3_ABL, never set on the product path): 1 no P / dS stores, 2 no dS ring (writes, flush reads),
// 4 no dropout hash, 8 V / Rd rows loaded once per tile only, 16 no exponentials
template <bool DROP, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void relattn_bwd_q3_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_Q3];
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
    LDS_AS char* const lds = (LDS_AS char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ii = lane & 31, half = lane >> 5, r16 = lane & 15;
    const int T = a.T, M = a.M, B = a.B, K = T + M, HD = a.H * 64;
    const int QT = (T + 127) / 128, QH = (QT + 1) / 2, QT64 = (T + 63) / 64;
    int qslot, h, b;
    tile_coords(QH, a.H, B, qslot, h, b);
    const bool rst = a.reset != nullptr && a.reset[b] != 0;
    const unsigned rsb = (unsigned)B * a.ld_qkv * 2u, rdb = (unsigned)a.ld_rd * 2u;
    const float dsc = DROP ? a.drop_scale : 1.f;
    const size_t kvbytes = ((size_t)(K - 1) * B * a.ld_qkv + 64) * 2;
    const srd_t srdK = make_srd(a.k + (size_t)b * a.ld_qkv + h * 64, kvbytes);
    const srd_t srdV = make_srd(a.v + (size_t)b * a.ld_qkv + h * 64, kvbytes);
    const srd_t srdR = make_srd(a.rd + h * 64, ((size_t)(K - 1) * a.ld_rd + 64) * 2);
    const size_t bh = (size_t)b * a.H + h;

    // LDS-DMA of a 64-row K tile by 4 waves: piece j of wave w = rows 16 w + 8 j + (lane >> 3), 1 KB per instruction
    unsigned dvoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int drow = 16 * w + 8 * j + (lane >> 3);
        dvoff[j] = (unsigned)drow * rsb + (unsigned)(((lane & 7) ^ swz3(drow)) * 16);
    }
    const unsigned ldsw = lds0 + (unsigned)(OFF_K + w * 2048);

    // K rows as the A operand of S^T (32 keys x 16 k): accumulator row rho is key rho ^ 3 (see relattn3.hip: the ring stores
    // keys in descending column order; reversing the keys inside each quad makes the 16-byte ring reads land in register order)
    int fk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fk[ks] = (ii ^ 3) * 128 + (((2 * ks + half) ^ swz3(ii ^ 3)) << 4);
    // K^T as the A operand of dq^T (32 features x 16 keys) by transpose reads, rows reversed the same way
    int ka[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int R = 4 * half + 8 * X + 3 - (r16 >> 2), col = 32 * dt + 16 * ((lane >> 4) & 1) + 4 * (r16 & 3);
            ka[dt][X] = R * 128 + ((((col >> 3) ^ swz3(R))) << 4) + (col & 7) * 2;
        }
    // V / Rd rows as A operands straight from global memory: lane = row, 16 bytes at feature 16 ks + 8 half
    const unsigned fvoff = (unsigned)(ii ^ 3) * rsb + (unsigned)(16 * half);          // (+ 32 ks, + jb * rsb)
    const unsigned froff = (unsigned)(16 * half);                                     // (+ 32 ks, + d * rdb)

    // the wave's BD ring (relattn3.hip): row i holds key j at column (j - 4 (i >> 2)) mod 64, mirrored
    const int m4 = ii & 3;
    const int ringb = OFF_RING + w * RINGB;
    const int rowb = ringb + 4 * (68 * ii + 4 * (ii >> 2));
    const int c0 = 63 - 4 * half + m4;
    const int aw = rowb + 4 * (63 - c0);
    int aw0[3];
#pragma unroll
    for (int dr = 0; dr < 3; ++dr) aw0[dr] = rowb + 4 * (63 - ((c0 - dr) & 63));
    int ar[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) ar[u][q] = rowb + 4 * (60 - ((32 * u + 8 * q + 4 * half - 4 * (ii >> 2)) & 63));
    // the wave's dS-by-distance ring: row ii at 128 ii, distance d at column d & 63
    const unsigned dsrow = lds0 + (unsigned)(OFF_DS + w * DSB + ii * 128);
    const int dsflush = OFF_DS + w * DSB + (lane >> 1) * 128;          // flush: lane (row = lane >> 1, k = lane & 1)

    const unsigned key_bh = DROP ? mix32(salted(a.drop_seed) + (unsigned)(b * a.H + h) * 0x9E3779B1u) : 0u;
    const unsigned xl = (unsigned)(((ii >> 1) << 4) | (2 * half)) * DROP_C1;
    const bool iodd = (ii & 1) != 0;
    const unsigned cme = iodd ? DROP_CM[1][0] : DROP_CM[0][0], cmo = iodd ? DROP_CM[1][1] : DROP_CM[0][1];
    const unsigned thr32 = a.drop_thr << 16;

    auto tr8 = [&](int byte_off) {
        return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(lds + byte_off)));
    };
    const size_t mrow0 = (size_t)T * B;
    const srd_t srdD = make_srd(a.dsk + (size_t)h * mrow0 * a.ld_dsk, mrow0 * a.ld_dsk * 2);
    const int KS32 = 2 * ((K + 63) >> 6), QB32 = (T + 31) >> 5;

    for (int rep = 0; rep < 2; ++rep) {
        const int qt = rep == 0 ? QT - 1 - qslot : qslot;
        if (rep == 1 && qt >= QT - 1 - qslot) break;              // odd tile count: the middle tile is done once
        const int i0 = qt * 128, iw = i0 + 32 * w;
        int jt_lo, jt_hi;
        kv_range(a, i0, 128, rst, jt_lo, jt_hi);
        const int NT = jt_hi - jt_lo + 1;
        const int E = i0 + M - 64 * jt_lo - 63;                   // block m of wave w: distances E + 32 (w - m + 1) .. + 31
        const bool active = iw < T;
        const int jhi_w = min(K - 1, min(iw + 31, T - 1) + M);
        const int nsub_w = active ? ((jhi_w - 64 * jt_lo) >> 5) + 1 : 0;      // sub-tiles of 32 keys this wave computes
        if (rep == 1) __syncthreads();                            // the first tile's buffers and rings are free

        auto stage_k = [&](int t) {
            const unsigned off = (unsigned)((jt_lo + t) * 64) * rsb;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                lds_dma16s(srdK, dvoff[j], off, ldsw + (unsigned)((t & 1) * TILEB + j * 1024));
        };
        auto load_v = [&](bf16x8 (&f)[4], int n) {                // V rows of sub-tile n (rows beyond K: zeros)
            const unsigned off = (unsigned)(64 * jt_lo + 32 * n) * rsb + fvoff;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) f[ks] = buf_ld(srdV, off + 32u * ks);
        };
        auto load_r = [&](bf16x8 (&f)[4], int m) {                // Rd rows of band block m (distances outside [0, K): zeros)
            const unsigned off = (unsigned)(E + 32 * (w - m + 1) + ii) * rdb + froff;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) f[ks] = buf_ld(srdR, off + 32u * ks);
        };

        stage_k(0);
        // query-side operands (B operands: lane = query, 8 consecutive features), row statistics
        bf16x8 qu[4], qv[4], dof[4];
        float nls, ndl;
        {
            const int iq = min(iw + ii, T - 1);
            const size_t off = ((size_t)iq * B + b) * HD + h * 64 + 8 * half;
            const bf16* dop = a.dout + ((size_t)iq * B + b) * a.ld_o + h * 64 + 8 * half;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qu[ks] = ld_bf16x8(a.qu2 + off + 16 * ks);
                qv[ks] = ld_bf16x8(a.qv2 + off + 16 * ks);
                dof[ks] = ld_bf16x8(dop + 16 * ks);
            }
            // -lse2 and -delta/dsc are the INITIAL values of the band / dP accumulators: no subtraction per element
            nls = __log2f(a.scale * dsc) - a.lse_in[bh * T + iq] * LOG2E;
            ndl = -a.delta[bh * T + iq] / dsc;
        }
        bf16x8 vf[4], rf[4], rf0[4], rf1[4];
        load_r(rf0, -1);
        load_r(rf1, 0);
        load_v(vf, 0);
        load_r(rf, 1);
        // dS ring: everything beyond what a sub-tile writes must read as zero (distances right of the causal edge)
        {
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int n = 0; n < 4; ++n) *(LDS_AS bf16x8*)(lds + OFF_DS + w * DSB + (lane + 64 * n) * 16) = z;
        }
        // flush addressing of this tile: lane (row = lane >> 1, k = lane & 1) -> byte offset of its row in this head's
        // dS-by-distance block
        const int fl_i = iw + (lane >> 1);
        unsigned fl_row;
        {
            const unsigned m = (ABL & 512) ? (unsigned)b * (unsigned)T + (unsigned)min(fl_i, T - 1)          // (experiment: batch-major tiles)
                                           : (unsigned)min(fl_i, T - 1) * (unsigned)B + (unsigned)b;
            fl_row = a.dsk_tiled ? (((m >> 6) * (unsigned)(a.ld_dsk >> 7)) << 14) + ((m & 63u) << 8) : m * (unsigned)a.ld_dsk * 2u;
        }
        auto dsk_off = [&](int c) -> unsigned {                   // byte offset of aligned chunk c (distances 8c .. 8c+7) of the lane's row
            return a.dsk_tiled ? fl_row + ((((unsigned)(c >> 4)) << 13) + (unsigned)((8 * c) & 127)) * 2u
                               : fl_row + (unsigned)(8 * c) * 2u;
        };
        if (rst && jt_lo > 0) {
            // a sequence that starts here (reset_mems) skips its memory tiles: their distances (dtop, i + M] must still read as
            // zero for the band consumers (the scratch is re-used between layers and steps), so they are written
            const int dtop = fl_i + M - 64 * jt_lo, clast = (fl_i + M) >> 3;
            const u32x4 z4 = {0u, 0u, 0u, 0u};
            for (int c = (dtop >> 3) + 1 + (lane & 1); c <= ((iw + 31 + M) >> 3); c += 2) {
                unsigned off = dsk_off(c);
                if (!(fl_i < T && c <= clast)) off = 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(z4, srdD, (int)off, 0, 0);
            }
        }
        // P blocks of this wave's 32 rows: block (iw >> 5, ks32) at ((iw >> 5) KS32 + ks32) 2 KB of this (batch, head)'s part
        const srd_t srdP = make_srd((const char*)a.pbuf + ((bh * QB32 + (size_t)(iw >> 5)) * KS32) * 2048,
                                    (a.pbuf != nullptr && active) ? (size_t)KS32 * 2048 : 0);

        f32x16 dq[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

        auto band_write = [&](const f32x16& acc, int form, int r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (form == 0 && dr < 3) *(LDS_AS float*)(lds + aw0[dr]) = acc[r];
            else *(LDS_AS float*)(lds + aw + 4 * dr + (form == 0 ? 0 : 128)) = acc[r];
        };
        auto band_block = [&](const bf16x8 (&f)[4], int form) {   // (q+v) . Rd^T of one block -> ring, -lse2 folded in
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = nls;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = mfma32(f[ks], qv[ks], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) band_write(acc, form, r);
        };
        auto need_mask = [&](int n) {
            const int jb = 64 * jt_lo + 32 * n;
            return (jb + 31 > iw + M) || (a.same_length && jb <= iw + 31 - a.sshift) || (rst && jb < M) || (iw + 31 >= T);
        };

        // one sub-tile n (u = n & 1) of K tile buffer kb.  Order (pinned by scheduling barriers: the register budget is 256 and
        // a spill costs a scratch access that waits, in order, behind every store in flight): dP^T, S^T, element-wise + stores,
        // dq^T, then the band block of the NEXT sub-tile -- its accumulators are alive only while S and dP are dead
        auto step = [&](auto UC, int n, int kb) {
            constexpr int u = decltype(UC)::value;
            const int jb = 64 * jt_lo + 32 * n;
            const int kt = OFF_K + kb * TILEB + 4096 * u;
            f32x16 S, dp;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                          // BD (+ -lse2) from the ring = initial value of the score product
                const f32x4 v = *(const LDS_AS f32x4*)(lds + ar[u][q]);
                S[4 * q + 0] = v[0]; S[4 * q + 1] = v[1]; S[4 * q + 2] = v[2]; S[4 * q + 3] = v[3];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = ndl;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp = mfma32(vf[ks], dof[ks], dp);                       // dP^T - delta/dsc
            if (!(ABL & (8 | 128))) load_v(vf, n + 1);                     // the next step's V rows (after their last use above)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) S = mfma32(*(const LDS_AS bf16x8*)(lds + kt + fk[ks]), qu[ks], S);
            __builtin_amdgcn_sched_barrier(0);
            if (need_mask(n)) {
                const int i = iw + ii;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (is_masked(i, jb + 3 - (r & 3) + 8 * (r >> 2) + 4 * half, M, a.same_length, a.sshift, rst) || i >= T)
                        S[r] = -INFINITY;
            }
            unsigned k1 = 0, k2 = 0, k3 = 0;
            if (DROP) {
                k1 = mix32k(((unsigned)(iw >> 5) << 16) | (unsigned)(2 * jt_lo + n), key_bh);
                k2 = iodd ? k1 * DROP_KA[1][0] + DROP_KB[1][0] : k1 * DROP_KA[0][0] + DROP_KB[0][0];
                k3 = iodd ? k1 * DROP_KA[1][1] + DROP_KB[1][1] : k1 * DROP_KA[0][1] + DROP_KB[0][1];
            }
            unsigned pw[8], dw[8];                                 // P scale/(1-p) (sign = dropped) and dS, bf16 pairs
            // byte column of key jb + 4 half + 3 (register 0) in the dS ring, before the wrap
            const int dcol2 = 2 * (iw + ii + M - jb - 4 * half - 3);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float p0 = (ABL & 16) ? S[2 * k] : __builtin_amdgcn_exp2f(S[2 * k]), p1 = (ABL & 16) ? S[2 * k + 1] : __builtin_amdgcn_exp2f(S[2 * k + 1]);
                float s0 = p0, s1 = p1, e0 = dp[2 * k], e1 = dp[2 * k + 1];
                if (DROP && !(ABL & 4)) {
                    // registers 2k, 2k+1 hold the odd and the even key of cell 4 (k >> 1) + 2 half + 1 - (k & 1) (relattn3.hip)
                    unsigned y = xl + ((unsigned)(4 * (k >> 1) + 1 - (k & 1)) * DROP_C1 + k1);
                    y ^= y >> 12;
                    const unsigned w0 = (y & 0xFFFFFFu) * cme + k2, w1 = (y & 0xFFFFFFu) * cmo + k3;
                    const bool keep0 = w1 >= thr32, keep1 = w0 >= thr32;
                    s0 = keep0 ? p0 : -p0;
                    s1 = keep1 ? p1 : -p1;
                    e0 = keep0 ? e0 : ndl;                         // dropped: keep * dP = 0, the -delta term stays
                    e1 = keep1 ? e1 : ndl;
                }
                pw[k] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){s0, s1}, bf16x2));
                const bf16x2 d2 = __builtin_convertvector((f32x2){p0 * e0, p1 * e1}, bf16x2);
                dw[k] = __builtin_bit_cast(unsigned, d2);
                // by distance: register r = 2k + e is key jb + 8 (r >> 2) + 4 half + 3 - (r & 3), distance i + M - key
#pragma unroll
                for (int e = 0; e < ((ABL & 2) ? 0 : 2); ++e) {
                    const int r = 2 * k + e;
                    const unsigned ad = ((unsigned)(dcol2 + 2 * (r & 3) - 16 * (r >> 2)) & 126u) | dsrow;
                    *(LDS_AS bf16*)(size_t)ad = d2[e];
                }
            }
            // P for the key-stationary kernel: piece k2 of lane l at 1024 k2 + 16 l of block (iw >> 5, 2 jt_lo + n)
            if (!(ABL & (1 | 32))) {
                const int so = (2 * jt_lo + n) << 11;
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){pw[0], pw[1], pw[2], pw[3]}, srdP, lane * 16, so, 2);
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){pw[4], pw[5], pw[6], pw[7]}, srdP, 1024 + lane * 16, so, 2);
            }
            __builtin_amdgcn_wave_barrier();
            // flush: the sub-tile completed distances [i + M - jb - 31, ..): the 4 aligned chunks from c0 = ceil(that / 8), two
            // lanes per row.  (Measured: ONE 64-byte-aligned segment of 32 distances per row and step from four lanes -- half the
            // write requests -- is SLOWER, 765 against 650 us per launch; so is nothing about the tile order: batch-major tiles,
            // where a wave's rows are neighbours in memory, measure the same.)
            if (!(ABL & 3)) {
                const int dlo = fl_i + M - jb - 31;
                const int cf = ((dlo + 7) >> 3) + (lane & 1);
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    const int c = cf + 2 * n2;
                    const bf16x8 v8 = *(const LDS_AS bf16x8*)(lds + dsflush + ((8 * c) & 63) * 2);
                    unsigned off = dsk_off(c);
                    if (!(fl_i < T && c >= 0)) off = 0x80000000u;
                    if (!(ABL & 64)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v8), srdD, (int)off, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
            // dq^T += K^T . dS^T
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x4 lo = tr8(kt + ka[dt][0] + 2048 * ks2), hi = tr8(kt + ka[dt][1] + 2048 * ks2);
                    bf16x8 kf;
                    kf[0] = lo[0]; kf[1] = lo[1]; kf[2] = lo[2]; kf[3] = lo[3]; kf[4] = hi[0]; kf[5] = hi[1]; kf[6] = hi[2]; kf[7] = hi[3];
                    const u32x4 dq4 = {dw[4 * ks2], dw[4 * ks2 + 1], dw[4 * ks2 + 2], dw[4 * ks2 + 3]};
                    dq[dt] = mfma32(kf, __builtin_bit_cast(bf16x8, dq4), dq[dt]);
                }
            __builtin_amdgcn_sched_barrier(0);
            band_block(rf, u);                                     // block n + 1 (form (n + 1) & 1 ? 0 : 1 = u): after this step's ring read
            if (!(ABL & (8 | 256))) load_r(rf, n + 2);
            __builtin_amdgcn_sched_barrier(0);
        };

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // K tile 0 has landed
        if (nsub_w > 0) {                                         // distances of sub-tile 0: two blocks
            band_block(rf0, 0);
            band_block(rf1, 1);
        }
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) stage_k(t + 1);                       // (every wave has passed the barrier that ended tile t - 1)
            if (2 * t < nsub_w) step(std::integral_constant<int, 0>{}, 2 * t, t & 1);
            if (2 * t + 1 < nsub_w) step(std::integral_constant<int, 1>{}, 2 * t + 1, t & 1);
            // the next K tile must have landed.  Its two DMA pieces were issued before this tile's steps, and a step issues
            // exactly 12 vector-memory instructions (4 V rows, 2 P stores, 2 flush stores, 4 Rd rows; predicated-off stores aim
            // out of range instead of branching): everything younger may stay in flight -- a vmcnt(0) here would wait for the
            // acknowledgement of the stores just issued, a round trip to memory per tile
            if (ABL != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (2 * t + 1 < nsub_w) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (a.dsk_wedge > 0 && active) {          // zeros right of the causal edge, as far as the band pass / GEMMs read
            for (int r = 0; r < 32; ++r) {
                const int i = iw + r;
                if (i >= T) break;
                const size_t m = (size_t)i * B + b;
                const int dbeg = i + M + 1, dend = min(a.ld_dsk, dbeg + a.dsk_wedge);
                for (int d = dbeg + lane; d < dend; d += 64) {
                    bf16* dst = a.dsk_tiled
                        ? a.dsk + ((((size_t)h * (mrow0 >> 6) + (m >> 6)) * (a.ld_dsk >> 7) + (d >> 7)) << 13) + ((m & 63) << 7) + (d & 127)
                        : a.dsk + ((size_t)h * mrow0 + m) * a.ld_dsk + d;
                    *dst = f2bf(0.f);
                }
            }
        }
        // column sums of dq (for d r_w_bias) per 64-row tile: dq^T (fp32) through the wave's BD ring area as [64 features][32
        // queries], every lane sums one feature row
        {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * dt + 8 * (r >> 2) + 4 * half + (r & 3);
                    *(LDS_AS float*)(lds + ringb + (f * 32 + ii) * 4) = dq[dt][r];
                }
            __builtin_amdgcn_wave_barrier();
            float cs = 0.f;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const f32x4 v = *(const LDS_AS f32x4*)(lds + ringb + lane * 128 + n * 16);
                cs += (v[0] + v[1]) + (v[2] + v[3]);
            }
            *(LDS_AS float*)(lds + OFF_RED + (w * 64 + lane) * 4) = cs;
            __builtin_amdgcn_wave_barrier();
        }
        // dq (AC part): dq^T through the wave's ring area as [32 rows][64] bf16 (chunk c of row r at c ^ (r & 7)), out as
        // whole 128-byte rows
        if (active) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4 ob;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ob[e] = f2bf(dq[dt][4 * q + e]);
                    *(LDS_AS bf16x4*)(lds + ringb + ii * 128 + (((4 * dt + q) ^ (ii & 7)) << 4) + 8 * half) = ob;
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int row = (lane >> 3) + 8 * n, ch = lane & 7, i = iw + row;
                const bf16x8 v = *(const LDS_AS bf16x8*)(lds + ringb + row * 128 + ((ch ^ (row & 7)) << 4));
                if (i < T) st_bf16x8(a.dq + ((size_t)i * B + b) * HD + h * 64 + 8 * ch, v);
            }
        }
        __syncthreads();
        if (tid < 128) {                                          // du_part rows are 64-query tiles: waves 0, 1 and 2, 3
            const int hs = tid >> 6, f = tid & 63, q64 = 2 * qt + hs;
            if (q64 < QT64)
                a.du_part[((size_t)b * QT64 + q64) * HD + h * 64 + f] =
                    *(const LDS_AS float*)(lds + OFF_RED + ((2 * hs) * 64 + f) * 4) + *(const LDS_AS float*)(lds + OFF_RED + ((2 * hs + 1) * 64 + f) * 4);
        }
    }
}

}  // namespace

// the shapes this kernel takes (the launcher in relattn.hip falls back to relattn_bwd_q_kernel otherwise)
bool relattn_bwd_q3_takes(const AttnArgs& a) {
    const size_t K = (size_t)a.T + a.M;
    return a.pbuf != nullptr && a.pf == nullptr && a.o_in == nullptr && (a.ld_dsk % 8) == 0 &&
           (size_t)(2 * ((K + 63) / 64)) * 2048 < 0x7FFF0000ull;
}

int launch_relattn_bwd_q3(const AttnArgs& a, hipStream_t stream) {
    const int QT = (a.T + 127) / 128;
    dim3 grid(((QT + 1) / 2) * a.H * a.B);
    static const int abl = getenv("COMMU_Q3_ABL") ? atoi(getenv("COMMU_Q3_ABL")) : 0;
    if (abl != 0) {          // (profiling only: wrong results by construction)
        switch (abl) {
            case 1: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 1>), grid, dim3(256), 0, stream, a); break;
            case 3: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 3>), grid, dim3(256), 0, stream, a); break;
            case 7: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 7>), grid, dim3(256), 0, stream, a); break;
            case 15: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 15>), grid, dim3(256), 0, stream, a); break;
            case 31: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 31>), grid, dim3(256), 0, stream, a); break;
            case 4: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 4>), grid, dim3(256), 0, stream, a); break;
            case 32: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 32>), grid, dim3(256), 0, stream, a); break;
            case 64: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 64>), grid, dim3(256), 0, stream, a); break;
            case 128: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 128>), grid, dim3(256), 0, stream, a); break;
            case 256: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 256>), grid, dim3(256), 0, stream, a); break;
            case 512: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 512>), grid, dim3(256), 0, stream, a); break;
            case 8: COMMU_LAUNCH((relattn_bwd_q3_kernel<true, 8>), grid, dim3(256), 0, stream, a); break;
            default: return -22;
        }
        COMMU_LAUNCH_CHECK();
        return 0;
    }
    if (a.drop_thr != 0u) COMMU_LAUNCH((relattn_bwd_q3_kernel<true>), grid, dim3(256), 0, stream, a);
    else COMMU_LAUNCH((relattn_bwd_q3_kernel<false>), grid, dim3(256), 0, stream, a);
    COMMU_LAUNCH_CHECK();
    return 0;
}
