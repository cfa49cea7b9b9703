// Key-stationary attention backward (dk, dv) from STORED probabilities on the 32x32 MFMA / transposed-role layout (gfx950,
// d_head 64).  The counterpart of relattn3.hip for the second backward kernel: relattn_bwd_kv2_kernel (relattn.hip) gives
// each wave 16 keys, so the dO / (q+u) tiles of a 64-query step are fetched from LDS as MFMA operands once per 16 keys; here a
// wave owns 32 keys on v_mfma_f32_32x32x16_bf16 and the lane is the KEY:
//   dP   [q, key] = dO . V^T          A = dO rows (ds_read_b128), B = V^T from registers (the wave's 32 keys, stationary);
//                                     C layout: lane = key, registers = 16 queries -- the layout P is stored in
//   dV^T [key, d] += Pm^T . dO        A = the lane's 16 probabilities AS THEY STAND in the registers (the contraction index q
//   dK^T [key, d] += dS^T . (q+u)         runs over the accumulator rows: a k-permutation, applied to B as well), B = dO^T /
//                                     (q+u)^T by ds_read_b64_tr_b16 with the same row permutation (relattn3.hip's V^T reads)
// P arrives from relattn_bwd_q_kernel (AttnArgs.p_layout 1) as 2-KB blocks of [32 queries x 32 keys] in exactly the
// accumulator order of this kernel -- lane (key, half) reads its 16 values as 32 contiguous bytes, a wave 2 KB -- with the
// dropout keep decision in the sign.  Math and constants: relattn_bwd_kv2_kernel.
// Reference: autograd of commu/model/model.py:313-345 (dk, dv terms).
#include "relattn_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// 16-byte chunk swizzle of a [64 rows][64 features] tile (relattn3.hip): bits 1..3 of the row, bit-reversed
__device__ __forceinline__ int swz3(int R) {
    const int p = R >> 1;
    return ((p & 1) << 2) | (p & 2) | ((p >> 2) & 1);
}
__device__ __forceinline__ void lds_dma4(srd_t srd, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(srd) : "memory");
}

constexpr int TILEB = 8192;          // one [64 q][64 f] bf16 tile

// P2 (AttnArgs.p_layout 2, written by relattn_bwd_q3_kernel): a block holds [32 queries x 32 keys] in the QUERY-stationary
// kernel's accumulator order -- lane (query, half) of that kernel wrote its registers 8k .. 8k+7 (keys 16k + 8qq + 4 half + 3 - e) as
// the 16 bytes at 1024 k + 16 (query + 32 half).  Here the lane is the KEY: a wave brings its block to a private 2-KB LDS image by
// two LDS-DMA instructions (slot 4 query + 2 half + k, so that the reads below are conflict-free) and fetches the 16 queries of
// its key with four ds_read_b64_tr_b16; the transpose hands lane l the key (l & 31) ^ 3 of the block (the quads are stored
// reversed), so the V^T fragments and the dk / dv rows use that key order too.
// PL = AttnArgs.p_layout.  0: the block order of the 16x16 pair (relattn_bwd_q_kernel's whole-line stores; four 8-byte loads per
// lane and 32x32 block) -- the default pairing since round 6: with p_layout 1 the query-stationary kernel's 8-byte P pieces cost it
// +0.23 GB of written bytes per launch (WRITE_SIZE: 1.47 against 1.24 GB), which ate the gain of this kernel inside the step.
template <bool DROP, int PL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void relattn_bwd_kv3_kernel(const AttnArgs a) {
    constexpr bool P2 = PL == 2;
    __shared__ __attribute__((aligned(1024))) char smem[4 * TILEB + 2 * 256 + (P2 ? 4 * 4096 : 0)];          // dO x2, (q+u) x2, delta x2, P blocks
    constexpr int OFF_O = 0, OFF_Q = 2 * TILEB, OFF_D = 4 * TILEB, OFF_P = 4 * TILEB + 512;
    const LDS_AS char* lds = (const LDS_AS char*)smem;
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ii = lane & 31, half = lane >> 5, r16 = lane & 15;
    const int T = a.T, M = a.M, B = a.B, K = T + M, HD = a.H * 64;
    const int NT = (K + 127) >> 7, NH = (NT + 1) >> 1;          // key tiles of 128 (32 per wave), tiles j and NT-1-j back to back
    const int JT = (K + 63) >> 6, KS32 = 2 * JT;
    int jslot, h, b;
    tile_coords(NH, a.H, B, jslot, h, b);
    const bool rst = a.reset != nullptr && a.reset[b] != 0;
    const float ndsc = DROP ? -1.f / a.drop_scale : -1.f;          // -1 / dsc
    // stored P scale/(1-p) -> P' = P ln2/(1-p): the factor is linear in both products and is applied to dk, dv at the end
    const float pmul = LN2 / a.scale;
    const size_t bh = (size_t)b * a.H + h;

    // LDS-DMA staging of a 64-row tile by 4 waves: piece j of wave w = rows 16 w + 8 j + (lane >> 3), 1 KB per instruction
    unsigned dvoffQ[2], dvoffO[2];
    const unsigned qsb = (unsigned)B * HD * 2u, osb = (unsigned)B * a.ld_o * 2u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int drow = 16 * w + 8 * j + (lane >> 3);
        const unsigned dchunk = (unsigned)(((lane & 7) ^ swz3(drow)) * 16);
        dvoffQ[j] = (unsigned)drow * qsb + dchunk;
        dvoffO[j] = (unsigned)drow * osb + dchunk;
    }
    const srd_t srdQu = make_srd(a.qu2 + (size_t)b * HD + h * 64, ((size_t)(T - 1) * B * HD + 64) * 2);
    const srd_t srdO = make_srd(a.dout + (size_t)b * a.ld_o + h * 64, ((size_t)(T - 1) * B * a.ld_o + 64) * 2);
    const srd_t srdDl = make_srd(a.delta + bh * T, (size_t)T * 4);
    auto stage = [&](int it, int buf) {          // rows >= T lie beyond the descriptors: zeros
        const unsigned i0 = (unsigned)it * 64u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            lds_dma16(srdO, i0 * osb + dvoffO[j], lds0 + OFF_O + buf * TILEB + (16 * w + 8 * j) * 128);
            lds_dma16(srdQu, i0 * qsb + dvoffQ[j], lds0 + OFF_Q + buf * TILEB + (16 * w + 8 * j) * 128);
        }
        if (w == 0) lds_dma4(srdDl, (i0 + (unsigned)lane) * 4u, lds0 + OFF_D + buf * 256);
    };

    // A operand (32 rows x 16 k) of a tile's 32-row half: row = lane & 31, chunk 2 ks + half
    int fa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa[ks] = ii * 128 + (((2 * ks + half) ^ swz3(ii)) << 4);
    // transposed B operand (16 k = query rows x 32 features) of a tile: features 32 dt + (lane & 31); k-slot (half, e) is
    // query row 16 t + 8 (e >> 2) + 4 half + (e & 3) of the 32-row half -- the accumulator row of register 8 t + e: two
    // ds_read_b64_tr_b16 (X = e >> 2), each lane supplying the address of 4 consecutive features of row 4 half + 8 X + (r16 >> 2)
    int va[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int R = 4 * half + 8 * X + (r16 >> 2), col = 32 * dt + 16 * ((lane >> 4) & 1) + 4 * (r16 & 3);
            va[dt][X] = R * 128 + (((col >> 3) ^ swz3(R)) << 4) + (col & 7) * 2;
        }
    auto tr8 = [&](int byte_off) {
        return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(lds + byte_off)));
    };
    auto trfrag = [&](int tile_off, int dt) {
        const bf16x4 lo = tr8(tile_off + va[dt][0]), hi = tr8(tile_off + va[dt][1]);
        bf16x8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    };

    for (int rep = 0; rep < 2; ++rep) {
        const int jt = rep == 0 ? jslot : NT - 1 - jslot;
        if (rep == 1 && jt <= jslot) break;
        const int j0 = jt * 128, jw = j0 + 32 * w;          // this wave's keys jw .. jw + 31
        const int jt64 = jw >> 6;                           // the 64-key tile (bwd_q's loop unit) they belong to
        // V^T fragments: lane key ii holds V[key][16 ks + 8 half .. + 8]
        bf16x8 vf[4];
        {
            const int j = jw + (P2 ? (ii ^ 3) : ii);
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            const size_t off = ((size_t)min(j, K - 1) * B + b) * a.ld_qkv + h * 64;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) vf[ks] = (j < K) ? ld_bf16x8(a.v + off + 16 * ks + 8 * half) : z;
        }
        f32x16 dv[2], dk[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dv[dt][r] = dk[dt][r] = 0.f;
        // query steps (64 rows) that see this key tile: i >= j - M; same_length: i < j + sshift
        int it_lo = max(0, j0 - M) >> 6;
        int it_hi = (T - 1) >> 6;
        if (a.same_length) it_hi = min(it_hi, (j0 + 127 + a.sshift - 1) >> 6);
        if (rst && j0 + 127 < M) it_hi = -1;          // whole tile is reset memory: no gradient
        if (it_hi < it_lo) it_hi = it_lo - 1;

        // P blocks of this wave: sub-tile (qs, ks32) at ((qs KS32 + ks32) 2 KB); the lane's 32 bytes at key ii, half
        // (a wave whose 32 keys lie beyond K -- the last 128-key tile of K % 128 <= 64 -- reads the last block column instead: unused,
        //  but a plain load past the end of the scratch faults)
        const bf16* pwave = a.pbuf + bh * (size_t)(2 * ((T + 31) >> 5)) * JT * 1024 + (size_t)min(jw >> 5, KS32 - 1) * 1024 + ii * 32 + half * 16;
        const int QS = (T + 31) >> 5;
        bf16x8 pn[2][2];
        // PL 0: blocks of [64 keys][16 rows] (2 KB, [B*H][ceil(T/16)][JT]); key k's 16 rows are 32 contiguous bytes, the four
        // 4-row slots s at physical slot s ^ ((k >> 2) & 3) (relattn.hip pt_off).  The lane's registers 8 blk + 4 x + e are rows
        // 16 blk + 8 x + 4 half + e of the 32-query block: slot 2 x + half of 16-row block blk
        const int QB16 = (T + 15) >> 4;
        const int k64 = (jw & 63) + ii, ksw = (k64 >> 2) & 3;
        const bf16* pwave0 = a.pbuf + (bh * (size_t)QB16 * JT + (size_t)min(jw >> 6, JT - 1)) * 1024 + k64 * 16;
        auto pfetch = [&](int it) {          // (clamped: a block past the end is somebody else's and is not used)
            if (PL == 0) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) {
                        const bf16* p = pwave0 + (size_t)min(2 * (2 * it + qb) + blk, QB16 - 1) * JT * 1024;
                        const bf16x4 lo = *(const bf16x4*)(p + ((half ^ ksw) << 2)), hi = *(const bf16x4*)(p + (((2 + half) ^ ksw) << 2));
                        pn[qb][blk] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                return;
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const bf16* p = pwave + (size_t)min(2 * it + qb, QS - 1) * KS32 * 1024;
                pn[qb][0] = ld_bf16x8(p);
                pn[qb][1] = ld_bf16x8(p + 8);
            }
        };
        // P2: block (qs, ks32) of this wave -> its LDS image (one per query half qb); lane l of DMA instruction j fills slot
        // 64 j + l = 4 query + 2 half + k from the block's piece 1024 k + 16 (query + 32 half)
        const srd_t srdPB = make_srd((const char*)a.pbuf + bh * (size_t)QS * KS32 * 2048, P2 ? (size_t)QS * KS32 * 2048 : 0);
        unsigned pvo[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int slot = 64 * j + lane;
            pvo[j] = (unsigned)((slot & 1) * 1024 + ((slot >> 2) + 32 * ((slot >> 1) & 1)) * 16);
        }
        auto pdma = [&](int it, int qb) {          // (a block past the end lies beyond the descriptor: zeros)
            const unsigned so = (unsigned)(((2 * it + qb) * KS32 + (jw >> 5)) << 11);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                lds_dma16s(srdPB, pvo[j], so, lds0 + (unsigned)(OFF_P + w * 4096 + qb * 2048 + j * 1024));
        };
        // transpose read of register quad q4 (queries 8 q4 + 4 half + 0..3) of the lane's key: the 16 lanes of a group supply
        // the addresses of 4 queries x 4 key quads; key quad G = (r16 & 3) + 4 ((lane >> 4) & 1) sits in piece k = G >> 2 of the
        // query-kernel lane (query, G & 1), at byte 8 ((G >> 1) & 1) of its 16
        int pta[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int rho = 8 * q4 + 4 * half + (r16 >> 2), G = (r16 & 3) + 4 * ((lane >> 4) & 1);
            pta[q4] = OFF_P + w * 4096 + (4 * rho + 2 * (G & 1) + (G >> 2)) * 16 + 8 * ((G >> 1) & 1);
        }
        if (it_lo <= it_hi) {
            stage(it_lo, 0);
            if (P2) { pdma(it_lo, 0); pdma(it_lo, 1); }
            else pfetch(it_lo);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int it = it_lo; it <= it_hi; ++it) {
            const int buf = (it - it_lo) & 1, i0 = it * 64;
            bf16x8 px[2][2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (!P2) { px[qb][0] = pn[qb][0]; px[qb][1] = pn[qb][1]; }
            }
            if (it < it_hi) {
                stage(it + 1, buf ^ 1);
                if (!P2) pfetch(it + 1);
            }
            // the key tiles bwd_q visited (and stored P for) from this query step
            int jlo64, jhi64;
            kv_range(a, P2 ? (i0 & ~127) : i0, P2 ? 128 : 64, rst, jlo64, jhi64);
            const bool seen = jw < K && jt64 >= jlo64 && jt64 <= jhi64;
            const int tO = OFF_O + buf * TILEB, tQ = OFF_Q + buf * TILEB;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const int iq = i0 + 32 * qb;
                // (P2: the query-stationary kernel visits the 32-key sub-tiles up to the last key its 32 rows can see)
                const bool seen_qb = seen && iq < T && (!P2 || jw <= min(K - 1, min(iq + 31, T - 1) + M));
                if (P2) {
                    if (seen_qb) {
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const bf16x4 v4 = tr8(pta[q4] + qb * 2048);
                            px[qb][q4 >> 1][4 * (q4 & 1) + 0] = v4[0]; px[qb][q4 >> 1][4 * (q4 & 1) + 1] = v4[1];
                            px[qb][q4 >> 1][4 * (q4 & 1) + 2] = v4[2]; px[qb][q4 >> 1][4 * (q4 & 1) + 3] = v4[3];
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    if (it < it_hi) pdma(it + 1, qb);          // (this half's image is free again: single-buffered per wave)
                }
                if (!seen_qb) continue;
                // -delta / dsc as the initial value of the dP accumulator (C layout: register r is query 8 (r >> 2) + 4 half + (r & 3))
                f32x16 ndl;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 d4 = *(const LDS_AS f32x4*)(lds + OFF_D + buf * 256 + (32 * qb + 8 * q4 + 4 * half) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ndl[4 * q4 + e] = d4[e] * ndsc;
                }
                f32x16 dp = ndl;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    dp = mfma32(*(const LDS_AS bf16x8*)(lds + tO + 32 * qb * 128 + fa[ks]), vf[ks], dp);
                // rows of 16-row blocks that lie entirely beyond T were never stored: garbage -> zero (NaN-safe select)
                const int vrows = ((T + 15) & ~15) - iq;          // valid rows of this 32-row half (>= 16)
                bf16x8 pa[2], da[2];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = bf2f(px[qb][r >> 3][r & 7]);          // +-P scale/(1-p), negative: dropped (bwd_q)
                    if (!P2 && vrows < 32 && 8 * (r >> 2) + 4 * half + (r & 3) >= vrows) x = 0.f;          // (P2: whole blocks are written, rows beyond T as zeros)
                    const float p = DROP ? __builtin_fabsf(x) : x;
                    float pd = p, dpe = dp[r];
                    if (DROP) {
                        pd = __builtin_fmaxf(x, 0.f);
                        dpe = x > 0.f ? dpe : ndl[r];          // dropped: keep * dP = 0, the -delta term stays
                    }
                    pa[r >> 3][r & 7] = f2bf(pd);
                    da[r >> 3][r & 7] = f2bf(p * dpe);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const bf16x8 fo = trfrag(tO + (32 * qb + 16 * t) * 128, dt);
                        const bf16x8 fq = trfrag(tQ + (32 * qb + 16 * t) * 128, dt);
                        dv[dt] = mfma32(pa[t], fo, dv[dt]);
                        dk[dt] = mfma32(da[t], fq, dk[dt]);
                    }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // accumulators: column = feature 32 dt + ii, register r = key jw + (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jw + (P2 ? 3 - (r & 3) : (r & 3)) + 8 * (r >> 2) + 4 * half;
            if (j < K) {
                const size_t off = ((size_t)j * B + b) * a.ld_dqkv + h * 64 + ii;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    a.dk[off + 32 * dt] = f2bf(dk[dt][r] * pmul);
                    a.dv[off + 32 * dt] = f2bf(dv[dt][r] * (pmul / LN2));
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

int launch_relattn_bwd_kv3(const AttnArgs& a, hipStream_t stream) {
    const int K = a.T + a.M, NT = (K + 127) / 128;
    const dim3 grid(((NT + 1) / 2) * a.H * a.B);
    if (a.p_layout == 2) {
        if (a.drop_thr != 0u) COMMU_LAUNCH((relattn_bwd_kv3_kernel<true, 2>), grid, dim3(256), 0, stream, a);
        else COMMU_LAUNCH((relattn_bwd_kv3_kernel<false, 2>), grid, dim3(256), 0, stream, a);
        return 0;
    }
    if (a.p_layout == 0) {
        if (a.drop_thr != 0u) COMMU_LAUNCH((relattn_bwd_kv3_kernel<true, 0>), grid, dim3(256), 0, stream, a);
        else COMMU_LAUNCH((relattn_bwd_kv3_kernel<false, 0>), grid, dim3(256), 0, stream, a);
        return 0;
    }
    if (a.drop_thr != 0u) COMMU_LAUNCH((relattn_bwd_kv3_kernel<true, 1>), grid, dim3(256), 0, stream, a);
    else COMMU_LAUNCH((relattn_bwd_kv3_kernel<false, 1>), grid, dim3(256), 0, stream, a);
    return 0;
}
