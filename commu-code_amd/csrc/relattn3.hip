// K6, third generation: relative-position masked attention on the 32x32x16 MFMA with the score tile TRANSPOSED
// (gfx950, d_head 64).  Reference math: commu/model/model.py:313-345, _rel_shift :251-259, masks :549-574 (see
// relattn.hip for the distance-indexed restatement  S[i,j] = ((q_i+u).k_j + (q_i+v).Rd[i+M-j]) * scale).
//
// A wave owns 32 query rows and walks the keys in sub-tiles of 32.  It computes S^T = K . (q+u)^T, so that in the
// accumulator layout of the 32x32 MFMA (lane = column = QUERY, 16 registers = KEYS 8q + 4 half + e) a query's scores
// are lane-local: the row maximum is 15 in-lane max + one half-wave exchange, the row sum is lane-local, and the
// probabilities -- converted to bf16 in place -- ARE the B operand of O^T += V^T . P^T (the A operand V^T is fetched with
// transpose reads in the same key order).  No P round trip through LDS, no cross-lane reductions.
//
// Rel-shift.  The band product QR^T[d][i] = Rd[d] . (q_i+v) comes out of the MFMA indexed by DISTANCE; the score tile
// needs it indexed by KEY j = i + M - d -- a per-lane register index, i.e. not a register operation.  It goes through a
// per-wave fp32 ring in LDS, BD[32 rows][64 keys] (row i holds keys j mod 64, rotated by 4 (i >> 2) so that the four keys
// of an accumulator quad are one aligned 16-byte read while a quad of distances of the producer is four 4-byte writes at
// immediate offsets): every distance is computed ONCE (32 new distances per sub-tile: 4 MFMAs) and the quad reads land
// directly in the accumulator registers as the INITIAL value of the K . (q+u)^T product -- the skew costs no VALU work.
//
// Workgroup = 8 waves = 256 query rows of one (batch, head); K / V tiles of 64 keys and the distance table Rd (a ring of
// six 64-distance chunks: the eight waves' windows span 288 distances) arrive by LDS-DMA, double-buffered, one barrier per
// 64 keys.  Waves w and w + 4 share a SIMD and take row slices w and 7 - w, so every SIMD sees the same causal work.
#include "relattn_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max2f(float a, float b) {      // (fmaxf on an MFMA result costs a canonicalising v_max first)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max / sum of a value with the other half-wave's lane (lane ^ 32): v_permlane32_swap of two copies leaves {lo, lo} in one
// and {hi, hi} in the other.  (Inline asm: given the same value twice, hipcc folds the builtin's two results into one.)
__device__ __forceinline__ void xhalf_pair(float x, float& p, float& q) {
    p = x; q = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p), "+v"(q));
}
__device__ __forceinline__ float xhalf_max(float x) {
    float p, q;
    xhalf_pair(x, p, q);
    return max2f(p, q);
}
__device__ __forceinline__ float xhalf_sum(float x) {
    float p, q;
    xhalf_pair(x, p, q);
    return p + q;
}

constexpr int TILEB = 8192;                                   // [64 rows][64] bf16
constexpr int OFF_K = 0, OFF_V = 2 * TILEB, OFF_R = 5 * TILEB, NRCH = 6;      // K x2, V x3, Rd ring
constexpr int OFF_RING = OFF_R + NRCH * TILEB;
constexpr int RINGB = 8832;                                   // per wave: rows of 64 fp32 at word 68 i + 4 (i >> 2)
constexpr int LDS_FWD3 = OFF_RING + 8 * RINGB;                // 160 768 bytes

// 16-byte chunk c of row R of a [64][64] bf16 tile lives at chunk c ^ swz3(R): conflict-free for the ds_read_b128 of a
// 32-row MFMA operand (lanes = rows) and for ds_read_b64_tr_b16 (4 rows x 4 chunks per 32 lanes)
__device__ __forceinline__ int swz3(int R) {
    const int p = R >> 1;
    return ((p & 1) << 2) | (p & 2) | ((p >> 2) & 1);
}

// attention-probability dropout: the mask of relattn.hip (DropLane: one mixed word per 2 x 2 cell of a 32x32 block, one
// multiply-add per element with the constants of its place in the cell).  Here a lane is one query and an accumulator quad
// holds 4 consecutive keys: the two keys of a cell share the first round, the query's parity picks the lane's constants.
constexpr unsigned DROP_C1 = 0xD2B74Bu;
constexpr unsigned DROP_CM[2][2] = {{0x9E3779u, 0x85EBCBu}, {0xC2B2AFu, 0xB5297Bu}};
constexpr unsigned DROP_KA[2][2] = {{0x85EBCA6Bu, 0xC2B2AE35u}, {0x27D4EB2Fu, 0x165667B1u}};
constexpr unsigned DROP_KB[2][2] = {{0x6A09E667u, 0xBB67AE85u}, {0x3C6EF372u, 0xA54FF53Au}};

// SAVEP: the probabilities also go to a.pf for the backward pass (commu_relattn_fwd_save; see relattn_common.h)
template <bool DROP, bool SAVEP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void relattn_fwd3_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_FWD3];
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
    LDS_AS char* const lds = (LDS_AS char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (w >= 4) __builtin_amdgcn_s_setprio(1);                 // the later-dispatched half loses every arbitration otherwise
    const int ii = lane & 31, half = lane >> 5, r16 = lane & 15;
    const int s = w < 4 ? w : 11 - w;                          // row slice of this wave
    const int T = a.T, M = a.M, B = a.B, K = T + M;
    const int QT = (T + 255) / 256, QH = (QT + 1) / 2;
    int qslot, h, b;
    tile_coords(QH, a.H, B, qslot, h, b);
    const bool rst = a.reset != nullptr && a.reset[b] != 0;
    const unsigned rsb = (unsigned)B * a.ld_qkv * 2u, rdb = (unsigned)a.ld_rd * 2u;
    const float c2 = a.scale * LOG2E;
    const size_t kvbytes = ((size_t)(K - 1) * B * a.ld_qkv + 64) * 2;
    const srd_t srdK = make_srd(a.k + (size_t)b * a.ld_qkv + h * 64, kvbytes);
    const srd_t srdV = make_srd(a.v + (size_t)b * a.ld_qkv + h * 64, kvbytes);
    const srd_t srdR = make_srd(a.rd + h * 64, ((size_t)(K - 1) * a.ld_rd + 64) * 2);

    // LDS-DMA: wave w stages rows 8w .. 8w+7 of a 64-row tile (1 KB per instruction); lane l lands at byte 16 l of the
    // wave's slice = row 8w + (l >> 3), physical chunk l & 7, so it fetches source chunk (l & 7) ^ swz3(row)
    const int drow = 8 * w + (lane >> 3);
    const unsigned dchunk = (unsigned)(((lane & 7) ^ swz3(drow)) * 16);
    const unsigned ldsw = lds0 + (unsigned)w * 1024u;

    // A operand (32 rows x 16 k) from a 32-row half of a tile: row = lane & 31, chunk 2 ks + half
    int fa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa[ks] = ii * 128 + (((2 * ks + half) ^ swz3(ii)) << 4);
    // ... of the K tile: accumulator row rho is key rho ^ 3 (the ring stores keys in DESCENDING column order so that the
    // band producer's writes ascend with its registers; reversing the keys inside each quad makes the 16-byte ring reads
    // land in register order again, and the V^T transpose reads below supply their rows reversed the same way)
    int fk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fk[ks] = (ii ^ 3) * 128 + (((2 * ks + half) ^ swz3(ii ^ 3)) << 4);
    // V^T operand (32 features x 16 keys) by transpose reads: lane group (lane >> 4) & 1 covers features +16, each lane
    // supplies the address of 4 consecutive features of key row 4 half + 8 X + (r16 >> 2)  (+16 ks2, +32 u: immediates)
    int va[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int R = 4 * half + 8 * X + 3 - (r16 >> 2), col = 32 * dt + 16 * ((lane >> 4) & 1) + 4 * (r16 & 3);
            va[dt][X] = R * 128 + ((((col >> 3) ^ swz3(R))) << 4) + (col & 7) * 2;
        }
    // the wave's BD ring
    const int m4 = ii & 3;
    const int ringb = OFF_RING + w * RINGB;
    const int rowb = ringb + 4 * (68 * ii + 4 * (ii >> 2));
    const int c0 = 63 - 4 * half + m4;                         // un-mirrored column of distance offset 0 of a form-0 block: 59 .. 66
    // mirrored ring columns (63 - column): a block's register r (distance offset dr + 4 half, dr = (r & 3) + 8 (r >> 2))
    // goes to column 63 - c0 + dr of form 0, + 32 of form 1; form 0 wraps for dr < c0 - 63 (dr <= 2)
    const int aw = rowb + 4 * (63 - c0);
    int aw0[3];
#pragma unroll
    for (int dr = 0; dr < 3; ++dr) aw0[dr] = rowb + 4 * (63 - ((c0 - dr) & 63));
    int ar[2][4];                                              // quad q of a sub-tile of parity u: keys 8q + 4 half + 3 .. + 0
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) ar[u][q] = rowb + 4 * (60 - ((32 * u + 8 * q + 4 * half - 4 * (ii >> 2)) & 63));

    const unsigned key_bh = DROP ? mix32(salted(a.drop_seed) + (unsigned)(b * a.H + h) * 0x9E3779B1u) : 0u;
    const unsigned xl = (unsigned)(((ii >> 1) << 4) | (2 * half)) * DROP_C1;
    const bool iodd = (ii & 1) != 0;
    const unsigned cme = iodd ? DROP_CM[1][0] : DROP_CM[0][0], cmo = iodd ? DROP_CM[1][1] : DROP_CM[0][1];
    const unsigned thr32 = a.drop_thr << 16;

    auto tr8 = [&](int byte_off) {
        return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(lds + byte_off)));
    };

    for (int rep = 0; rep < 2; ++rep) {
        const int qt = rep == 0 ? QT - 1 - qslot : qslot;
        if (rep == 1 && qt >= QT - 1 - qslot) break;              // odd tile count: the middle tile is done once
        const int i0 = qt * 256, iw = i0 + 32 * s;
        int jt_lo, jt_hi;
        kv_range(a, i0, 256, rst, jt_lo, jt_hi);
        const int NT = jt_hi - jt_lo + 1;
        const int E = i0 + M - 64 * jt_lo - 63;                   // distance of row 0 of Rd chunk 0
        const bool active = iw < T;
        const int jhi_w = min(K - 1, min(iw + 31, T - 1) + M);
        const int nsub_w = active ? ((jhi_w - 64 * jt_lo) >> 5) + 1 : 0;      // sub-tiles of 32 keys this wave computes
        if (rep == 1) __syncthreads();                            // the first tile's buffers and rings are free

        auto stage_rd = [&](int c) {                              // chunk c: distances E + 64 c .. + 63 (zeros outside [0, K))
            const int slot = (c + 600) % NRCH;
            lds_dma16(srdR, (unsigned)(E + 64 * c + drow) * rdb + dchunk, ldsw + (unsigned)(OFF_R + slot * TILEB));
        };
        auto stage_kv = [&](int t) {                              // tile t of this pass: K double-, V triple-buffered
            const unsigned off = (unsigned)((jt_lo + t) * 64 + drow) * rsb + dchunk;
            lds_dma16(srdK, off, ldsw + (unsigned)(OFF_K + (t & 1) * TILEB));
            lds_dma16(srdV, off, ldsw + (unsigned)(OFF_V + (t % 3) * TILEB));
        };
        // the query rows and biases are requested BEFORE the tile DMAs: loads return in order, so their conversion below
        // runs while the 64 KB of tiles are still in flight instead of behind them
        bf16x8 qraw[4];
        f32x4 ubias[4][2], vbias[4][2];
        {
            const int iq = min(iw + ii, T - 1);
            const bf16* qp = a.q + ((size_t)iq * B + b) * a.ld_qkv + h * 64;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qraw[ks] = ld_bf16x8(qp + 16 * ks + 8 * half);
                const int f0 = h * 64 + 16 * ks + 8 * half;       // (r_w_bias / r_r_bias rows of 64 floats: 32-byte pieces)
                ubias[ks][0] = *(const f32x4*)(a.u + f0); ubias[ks][1] = *(const f32x4*)(a.u + f0 + 4);
                vbias[ks][0] = *(const f32x4*)(a.vb + f0); vbias[ks][1] = *(const f32x4*)(a.vb + f0 + 4);
            }
        }
        asm volatile("" ::: "memory");          // (keep the requests above the DMA issue)
#pragma unroll
        for (int c = -1; c <= 4; ++c) stage_rd(c);
        stage_kv(0);

        bf16x8 qu[4], qv[4];
        {
            const int irow = iw + ii;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 raw = qraw[ks];
                const f32x4 u0 = ubias[ks][0], u1 = ubias[ks][1], v0 = vbias[ks][0], v1 = vbias[ks][1];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = bf2f(raw[e]);
                    qu[ks][e] = f2bf((x + (e < 4 ? u0[e & 3] : u1[e & 3])) * c2);
                    qv[ks][e] = f2bf((x + (e < 4 ? v0[e & 3] : v1[e & 3])) * c2);
                }
            }
            (void)irow;
        }
        // probabilities for the backward pass (a.pf, see relattn_common.h): this wave's row of 2176-byte tiles, one per
        // 32-key sub-tile; an inactive wave or a forward-only call stores through an empty descriptor (dropped)
        const int NS32 = (K + 31) >> 5;
        const size_t pfrow = (((size_t)b * a.H + h) * (size_t)((T + 31) >> 5) + (size_t)(iw >> 5)) * (size_t)NS32;
        const srd_t srdPF = make_srd((const char*)a.pf + (SAVEP ? pfrow * PF_TILE_BYTES : 0),
                                     (SAVEP && active) ? (size_t)NS32 * PF_TILE_BYTES : 0);
        f32x16 O[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
        float mrow = -3.0e38f, lsA = 0.f, lsB = 0.f;
        int fak[4], vak[2][2];                                    // K / V fragment addresses of the current buffers
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fak[ks] = OFF_K + fk[ks];

        // ---- pieces of a sub-tile -----------------------------------------------------------------------------------
        // band block of sub-tile m: distances E + 32 (s - m + 1) .. + 31 against the wave's 32 queries -> BD ring; its ring
        // columns are (c0 - dr) for even m - 1 ("form 0") and (c0 - 32 - dr) for odd
        auto band_rbase = [&](int m) {
            const int hc = s - m + 1;
            return OFF_R + (((hc >> 1) + 600) % NRCH) * TILEB + (hc & 1) * 4096;
        };
        auto band_write = [&](const f32x16& acc, int form, int r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (form == 0 && dr < 3) *(LDS_AS float*)(lds + aw0[dr]) = acc[r];
            else *(LDS_AS float*)(lds + aw + 4 * dr + (form == 0 ? 0 : 128)) = acc[r];
        };
        auto band_block = [&](int m, int form) {                  // un-pipelined (prologue)
            const int rbase = band_rbase(m);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = mfma32(*(const LDS_AS bf16x8*)(lds + rbase + fa[ks]), qv[ks], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) band_write(acc, form, r);
        };
        auto ring_read = [&](f32x16& S, int par) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *(const LDS_AS f32x4*)(lds + ar[par][q]);
                S[4 * q + 0] = v[0]; S[4 * q + 1] = v[1]; S[4 * q + 2] = v[2]; S[4 * q + 3] = v[3];
            }
        };
        auto need_mask = [&](int m) {
            const int jb = 64 * jt_lo + 32 * m;
            return (jb + 31 > iw + M) || (a.same_length && jb <= iw + 31 - a.sshift) || (rst && jb < M);
        };
        auto apply_mask = [&](f32x16& S, int m) {
            const int jb = 64 * jt_lo + 32 * m, i = iw + ii;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (is_masked(i, jb + 3 - (r & 3) + 8 * (r >> 2) + 4 * half, M, a.same_length, a.sshift, rst)) S[r] = -INFINITY;
        };
        // running maximum (log2 domain) over the finished score tile S; O and the partial sums follow when it grows
        auto fold_max = [&](const f32x16& S) {
            float mx = max3f(S[0], S[1], S[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = max3f(mx, S[r], S[r + 1]);
            mx = max2f(mx, S[15]);
            mx = xhalf_max(mx);
            const float mnew = max2f(mrow, mx);
            if (__any(mnew > mrow)) {
                const float al = __builtin_amdgcn_exp2f(mrow - mnew);
                mrow = mnew;
                lsA *= al;
                lsB *= al;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[dt][r] *= al;
            }
        };

        // One pipelined step n (u = n & 1): softmax arithmetic of sub-tile n (scores Sc, complete and folded into mrow) in
        // eight slices of two keys, each slice behind one MFMA of sub-tile n+1's score product (ring values as the initial
        // accumulator) or of the band block of sub-tile n+2; then P(n) . V with the maximum of sub-tile n+1 folded in under
        // its MFMAs.  FULL: sub-tiles n+1 and n+2 exist (no tests inside).
        int rb_step = band_rbase(2);                              // Rd half-chunk of the block step n computes (sub-tile n + 2)
        auto rb_advance = [&]() {                                 // one half-chunk (32 distances) down, ring of NRCH chunks
            rb_step -= 4096;
            if (rb_step < OFF_R) rb_step += NRCH * TILEB;
        };
        auto step = [&](auto UC, auto FC, f32x16& Sc, f32x16& Sn, int n) {
            constexpr int u = decltype(UC)::value;
            constexpr bool FULL = decltype(FC)::value;
            const bool h1 = FULL || (n + 1 < nsub_w), h2 = FULL || (n + 2 < nsub_w);
            bf16x8 kf[4], rf[4];
            f32x16 acc;
            if (h1) {
                ring_read(Sn, u ^ 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const LDS_AS bf16x8*)(lds + fak[ks] + 4096 * (u ^ 1));
            }
            if (h2) {
                const int rbase = rb_step;                        // == band_rbase(n + 2), kept incrementally (tile loops)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) rf[ks] = *(const LDS_AS bf16x8*)(lds + rbase + fa[ks]);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
            unsigned k1 = 0, k2 = 0, k3 = 0;
            if (DROP) {
                k1 = mix32k(((unsigned)(iw >> 5) << 16) | (unsigned)(2 * jt_lo + n), key_bh);
                // (additive keys of the even / odd key of a cell, for this lane's query parity)
                k2 = iodd ? k1 * DROP_KA[1][0] + DROP_KB[1][0] : k1 * DROP_KA[0][0] + DROP_KB[0][0];
                k3 = iodd ? k1 * DROP_KA[1][1] + DROP_KB[1][1] : k1 * DROP_KA[0][1] + DROP_KB[0][1];
            }
            unsigned pw[8];
            unsigned ps[4];                                       // (DROP) signed copies for a.pf, four at a time
            const unsigned psoff = (unsigned)(2 * jt_lo + n) * (unsigned)PF_TILE_BYTES;
            // the running maximum this sub-tile is exponentiated against (constant over the step): the tile's last 128 bytes
            // (lanes of half 1 aim out of range: dropped)
            if (SAVEP)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mrow), srdPF,
                                                      half == 0 ? 2048 + ii * 4 : (int)0x80000000u, (int)psoff, 0);
            auto sm = [&](int k) {                                // keys 2k, 2k+1 of the quad layout: exp, sums, dropout, bf16 pair
                float e0 = __builtin_amdgcn_exp2f(Sc[2 * k] - mrow), e1 = __builtin_amdgcn_exp2f(Sc[2 * k + 1] - mrow);
                lsA += e0;                                        // the normaliser is the un-dropped sum
                lsB += e1;
                if (DROP) {                                       // 1/(1-p) is applied to O at the end
                    // registers 2k, 2k+1 hold keys 8 (k >> 1) + 4 half + 3 - 2 (k & 1) and the one below it: key pair
                    // 4 (k >> 1) + 2 half + 1 - (k & 1), odd key first
                    unsigned y = xl + ((unsigned)(4 * (k >> 1) + 1 - (k & 1)) * DROP_C1 + k1);
                    y ^= y >> 12;
                    const unsigned w0 = (y & 0xFFFFFFu) * cme + k2, w1 = (y & 0xFFFFFFu) * cmo + k3;
                    // saved for the backward pass BEFORE the mask is applied: a dropped probability keeps its value (dS needs
                    // it) and carries the decision in its sign
                    if (SAVEP)
                        ps[k & 3] = __builtin_bit_cast(unsigned, __builtin_convertvector(
                            (f32x2){w1 >= thr32 ? e0 : -e0, w0 >= thr32 ? e1 : -e1}, bf16x2));
                    e0 = w1 >= thr32 ? e0 : 0.f;
                    e1 = w0 >= thr32 ? e1 : 0.f;
                }
                // (the compiler's own v_cvt_pk_bf16_f32: it knows the wait state a transcendental result needs)
                pw[k] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){e0, e1}, bf16x2));
                if (SAVEP && (k & 3) == 3) {                      // registers 8 (k >> 2) .. + 7 of the lane: 16 of its 32 bytes
                    const u32x4 v4 = DROP ? (u32x4){ps[0], ps[1], ps[2], ps[3]} : (u32x4){pw[k - 3], pw[k - 2], pw[k - 1], pw[k]};
                    __builtin_amdgcn_raw_buffer_store_b128(v4, srdPF, lane * 32 + 16 * (k >> 2), (int)psoff, 0);
                }
            };
            bf16x4 vt[2][2][2];                                   // [ks2][dt][X]
            auto vread = [&](int dt) {
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                    for (int X = 0; X < 2; ++X) vt[ks2][dt][X] = tr8(vak[dt][X] + 2048 * ks2 + 4096 * u);
            };
            auto pv = [&](int ks2, int dt) {
                bf16x8 vf;
                const bf16x4 lo = vt[ks2][dt][0], hi = vt[ks2][dt][1];
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3]; vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                u32x4 pq;
                pq[0] = pw[4 * ks2]; pq[1] = pw[4 * ks2 + 1]; pq[2] = pw[4 * ks2 + 2]; pq[3] = pw[4 * ks2 + 3];
                O[dt] = mfma32(vf, __builtin_bit_cast(bf16x8, pq), O[dt]);
            };
#define SB() __builtin_amdgcn_sched_barrier(0)
            SB();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {          // (arithmetic first: the first slices run while the operand reads land)
                sm(2 * ks);
                if (h1) Sn = mfma32(kf[ks], qu[ks], Sn);
                if (ks == 1) vread(0);
                SB();
                sm(2 * ks + 1);
                if (h2) acc = mfma32(rf[ks], qv[ks], acc);
                if (ks == 1) vread(1);
                SB();
            }
            pv(0, 0);
            SB();
            if (!FULL && h1 && need_mask(n + 1)) apply_mask(Sn, n + 1);      // (a straight-line step never meets a mask)
            SB();
            pv(0, 1);
            if (h2) {
#pragma unroll
                for (int r = 0; r < 8; ++r) band_write(acc, u ^ 1, r);
            }
            SB();
            pv(1, 0);
            if (h2) {
#pragma unroll
                for (int r = 8; r < 16; ++r) band_write(acc, u ^ 1, r);
            }
            SB();
            pv(1, 1);
            if (h1) fold_max(Sn);
#undef SB
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (nsub_w > 0) {                                         // distances of sub-tile 0: two blocks
            band_block(-1, 0);
            band_block(0, 1);
        }
        __builtin_amdgcn_s_barrier();                             // chunk 4 (first blocks of the top waves) is free
        if (NT > 1) {
            stage_kv(1);
            stage_rd(-2);
        }
        f32x16 SA, SB_;
        if (nsub_w > 0) {                                         // scores of sub-tile 0, then the block of sub-tile 1 (it
            ring_read(SA, 0);                                     // overwrites ring columns sub-tile 0 has just read)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) SA = mfma32(*(const LDS_AS bf16x8*)(lds + fak[ks]), qu[ks], SA);
            if (nsub_w > 1) band_block(1, 0);
            if (need_mask(0)) apply_mask(SA, 0);
            fold_max(SA);
        }
        // barrier b_t sits between steps 2t and 2t+1: tile t+1 has landed (K of its first half is read by step 2t+1), tile
        // t+2 may overwrite K(t) (last read by step 2t), V(t-1) and Rd chunk 3-t (last read by step 2t).
        // Two loops over the tiles with the same barrier sequence: first the tiles whose steps all have two successors
        // (straight-line bodies), then the wave's last tiles with tests inside (one instance of each step body per loop,
        // so the accumulators stay where they are across iterations).
        // (masks: only the causal edge, i.e. the sub-tiles from m0 on, when there are no reset / same_length masks)
        const int m0 = max(0, ((iw + M - 64 * jt_lo - 31) >> 5) + 1);
        const int t_full = (rst || a.same_length) ? 0 : min(NT, max(0, min((nsub_w - 2) >> 1, (m0 - 1) >> 1)));
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int X = 0; X < 2; ++X) vak[dt][X] = OFF_V + va[dt][X];
        int vbuf = 0;                                             // V buffer of the current tile (t % 3)
        auto tile_end = [&]() {                                   // V fragments of the next tile: next of the three buffers
            const int d = vbuf == 2 ? -2 * TILEB : TILEB;
            vbuf = vbuf == 2 ? 0 : vbuf + 1;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int X = 0; X < 2; ++X) vak[dt][X] += d;
        };
        auto tile_mid = [&](int t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 2 < NT) {
                stage_kv(t + 2);
                stage_rd(-3 - t);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fak[ks] ^= TILEB;      // K fragments now come from tile t+1
        };
        for (int t = 0; t < t_full; ++t) {
            step(std::integral_constant<int, 0>{}, std::true_type{}, SA, SB_, 2 * t);
            rb_advance();
            tile_mid(t);
            step(std::integral_constant<int, 1>{}, std::true_type{}, SB_, SA, 2 * t + 1);
            rb_advance();
            tile_end();
        }
        for (int t = t_full; t < NT; ++t) {
            if (2 * t < nsub_w) step(std::integral_constant<int, 0>{}, std::false_type{}, SA, SB_, 2 * t);
            rb_advance();
            tile_mid(t);
            if (2 * t + 1 < nsub_w) step(std::integral_constant<int, 1>{}, std::false_type{}, SB_, SA, 2 * t + 1);
            rb_advance();
            tile_end();
        }
        // epilogue: normalise, O^T through the wave's ring area as [32 rows][64] bf16 (chunk c of row r at c ^ (r & 7)),
        // out as whole 128-byte rows; lse
        // the scaled query operands for the backward pass leave HERE, with the tile's other stores: issued in the prologue
        // they sat in front of its `s_waitcnt vmcnt(0)`, which then also waited for eight store acknowledgements per lane
        if (a.qu2 != nullptr && iw + ii < T) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const size_t off = ((size_t)(iw + ii) * B + b) * (a.H * 64) + h * 64 + 16 * ks + 8 * half;
                st_bf16x8(a.qu2 + off, qu[ks]);
                st_bf16x8(a.qv2 + off, qv[ks]);
            }
        }
        if (active) {
            const float l = xhalf_sum(lsA + lsB);
            const float inv = (DROP ? a.drop_scale : 1.f) / l;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4 ob;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ob[e] = f2bf(O[dt][4 * q + e] * inv);
                    *(LDS_AS bf16x4*)(lds + ringb + ii * 128 + (((4 * dt + q) ^ (ii & 7)) << 4) + 8 * half) = ob;
                }
            if (half == 0 && iw + ii < T) a.lse[((size_t)b * a.H + h) * T + iw + ii] = (mrow + __log2f(l)) * LN2;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int row = (lane >> 3) + 8 * n, ch = lane & 7, i = iw + row;
                const bf16x8 v = *(const LDS_AS bf16x8*)(lds + ringb + row * 128 + ((ch ^ (row & 7)) << 4));
                if (i < T) st_bf16x8(a.out + ((size_t)i * B + b) * a.ld_o + h * 64 + 8 * ch, v);
            }
        }
    }
}

}  // namespace

int launch_relattn_fwd3(const AttnArgs& a, hipStream_t stream) {
    const int QT = (a.T + 255) / 256;
    dim3 grid(((QT + 1) / 2) * a.H * a.B);
    if (a.pf != nullptr) {
        if (a.drop_thr != 0u) COMMU_LAUNCH((relattn_fwd3_kernel<true, true>), grid, dim3(512), 0, stream, a);
        else COMMU_LAUNCH((relattn_fwd3_kernel<false, true>), grid, dim3(512), 0, stream, a);
    } else {
        if (a.drop_thr != 0u) COMMU_LAUNCH((relattn_fwd3_kernel<true, false>), grid, dim3(512), 0, stream, a);
        else COMMU_LAUNCH((relattn_fwd3_kernel<false, false>), grid, dim3(512), 0, stream, a);
    }
    COMMU_LAUNCH_CHECK();
    return 0;
}

