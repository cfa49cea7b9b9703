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
constexpr int OFF_K = 0, OFF_V = 2 * TILEB, OFF_R = 4 * TILEB, NRCH = 6;
constexpr int OFF_RING = OFF_R + NRCH * TILEB;
constexpr int RINGB = 8832;                                   // per wave: rows of 64 fp32 at word 68 i + 4 (i >> 2)
constexpr int LDS_FWD3 = OFF_RING + 8 * RINGB;                // 152 576 bytes

// 16-byte chunk c of row R of a [64][64] bf16 tile lives at chunk c ^ swz3(R): conflict-free for the ds_read_b128 of a
// 32-row MFMA operand (lanes = rows) and for ds_read_b64_tr_b16 (4 rows x 4 chunks per 32 lanes)
__device__ __forceinline__ int swz3(int R) {
    const int p = R >> 1;
    return ((p & 1) << 2) | (p & 2) | ((p >> 2) & 1);
}

// attention-probability dropout, second form (pairs along the KEYS: the transposed layout holds 4 consecutive keys of one
// query per accumulator quad).  Per 32x32 block (i >> 5, j >> 5) of a (batch, head): scalar keys k1, k2, k3 from the
// strong hash; element (ii, jj): a = ((ii << 4 | jj >> 1) * C1 + k1), a ^= a >> 12, word = (a & 0xFFFFFF) * (jj & 1 ? C3 :
// C2) + (jj & 1 ? k3 : k2); keep = word >= thr16 << 16.  Host mirror: ops.attn_dropout_keep_mask(version=2).
constexpr unsigned DROP_C1 = 0xD2B74Bu, DROP_C2 = 0x9E3779u, DROP_C3 = 0x85EBCBu;

template <bool DROP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void relattn_fwd3_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_FWD3];
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
    LDS_AS char* const lds = (LDS_AS char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    // V^T operand (32 features x 16 keys) by transpose reads: lane group (lane >> 4) & 1 covers features +16, each lane
    // supplies the address of 4 consecutive features of key row 4 half + 8 X + (r16 >> 2)  (+16 ks2, +32 u: immediates)
    int va[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int R = 4 * half + 8 * X + (r16 >> 2), col = 32 * dt + 16 * ((lane >> 4) & 1) + 4 * (r16 & 3);
            va[dt][X] = R * 128 + ((((col >> 3) ^ swz3(R))) << 4) + (col & 7) * 2;
        }
    // the wave's BD ring
    const int m4 = ii & 3;
    const int ringb = OFF_RING + w * RINGB;
    const int rowb = ringb + 4 * (68 * ii + 4 * (ii >> 2));
    const int c0 = 63 - 4 * half + m4;                         // column of distance offset 0 of a parity-0 block: 59 .. 66
    const int aw = rowb + 4 * (c0 - 59);                       // + 4 (27 - dr) (+128 for parity 0), dr = (r & 3) + 8 (r >> 2)
    int aw0[3];
#pragma unroll
    for (int dr = 0; dr < 3; ++dr) aw0[dr] = rowb + 4 * ((c0 - dr) & 63);      // parity 0, dr < 3: may wrap
    int ar[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) ar[u][q] = rowb + 4 * ((32 * u + 8 * q + 4 * half - 4 * (ii >> 2)) & 63);

    const unsigned key_bh = DROP ? mix32(salted(a.drop_seed) + (unsigned)(b * a.H + h) * 0x9E3779B1u) : 0u;
    const unsigned xl = (unsigned)((ii << 4) | (2 * half)) * DROP_C1;
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
        auto stage_kv = [&](int jt, int buf) {
            const unsigned off = (unsigned)(jt * 64 + drow) * rsb + dchunk;
            lds_dma16(srdK, off, ldsw + (unsigned)(OFF_K + buf * TILEB));
            lds_dma16(srdV, off, ldsw + (unsigned)(OFF_V + buf * TILEB));
        };
#pragma unroll
        for (int c = -1; c <= 4; ++c) stage_rd(c);
        stage_kv(jt_lo, 0);

        bf16x8 qu[4], qv[4];
        {
            const int irow = iw + ii;
            const int iq = min(irow, T - 1);
            const bf16* qp = a.q + ((size_t)iq * B + b) * a.ld_qkv + h * 64;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 raw = ld_bf16x8(qp + 16 * ks + 8 * half);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int f = h * 64 + 16 * ks + 8 * half + e;
                    const float x = bf2f(raw[e]);
                    qu[ks][e] = f2bf((x + a.u[f]) * c2);
                    qv[ks][e] = f2bf((x + a.vb[f]) * c2);
                }
                if (a.qu2 != nullptr && irow < T) {
                    const size_t off = ((size_t)irow * B + b) * (a.H * 64) + h * 64 + 16 * ks + 8 * half;
                    st_bf16x8(a.qu2 + off, qu[ks]);
                    st_bf16x8(a.qv2 + off, qv[ks]);
                }
            }
        }
        f32x16 O[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
        float mrow = -3.0e38f, lsum = 0.f;
        int fak[4], vak[2][2];                                    // K / V fragment addresses of the current buffer
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fak[ks] = OFF_K + fa[ks];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int X = 0; X < 2; ++X) vak[dt][X] = OFF_V + va[dt][X];

        // band block hc: distances E + 32 hc .. + 31 against the wave's 32 queries -> BD ring (key-indexed)
        auto band_block = [&](int hc, int u) {
            const int chunk = hc >> 1;
            const int rbase = OFF_R + ((chunk + 600) % NRCH) * TILEB + (hc & 1) * 4096;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = mfma32(*(const LDS_AS bf16x8*)(lds + rbase + fa[ks]), qv[ks], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (u == 0 && dr < 3) *(LDS_AS float*)(lds + aw0[dr]) = acc[r];
                else *(LDS_AS float*)(lds + aw + 4 * (27 - dr) + (u == 0 ? 128 : 0)) = acc[r];
            }
        };

        auto subtile = [&](int t, int u) {
            const int jb = 64 * (jt_lo + t) + 32 * u;
            f32x16 S;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *(const LDS_AS f32x4*)(lds + ar[u][q]);
                S[4 * q + 0] = v[0]; S[4 * q + 1] = v[1]; S[4 * q + 2] = v[2]; S[4 * q + 3] = v[3];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                S = mfma32(*(const LDS_AS bf16x8*)(lds + fak[ks] + 4096 * u), qu[ks], S);
            const bool need_mask = (jb + 31 > iw + M) || (a.same_length && jb <= iw + 31 - a.sshift) || (rst && jb < M);
            if (need_mask) {
                const int i = iw + ii;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (is_masked(i, jb + (r & 3) + 8 * (r >> 2) + 4 * half, M, a.same_length, a.sshift, rst)) S[r] = -INFINITY;
            }
            // online softmax in the log2 domain; the other half-wave holds the other 16 keys of the same query
            float mx = max3f(S[0], S[1], S[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = max3f(mx, S[r], S[r + 1]);
            mx = max2f(mx, S[15]);
            mx = xhalf_max(mx);
            const float mnew = max2f(mrow, mx);
            if (__any(mnew > mrow)) {
                const float al = __builtin_amdgcn_exp2f(mrow - mnew);
                mrow = mnew;
                lsum *= al;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[dt][r] *= al;
            }
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(S[r] - mrow);
                lsum += p[r];                                     // the normaliser is the un-dropped sum
            }
            if (DROP) {                                           // 1/(1-p) is applied to O at the end
                const unsigned k1 = mix32k(((unsigned)(iw >> 5) << 16) | (unsigned)(jb >> 5), key_bh);
                const unsigned k2 = k1 * 0x85EBCA6Bu + 0x6A09E667u, k3 = k1 * 0xC2B2AE35u + 0xBB67AE85u;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        unsigned y = xl + ((unsigned)(4 * q + pr) * DROP_C1 + k1);
                        y ^= y >> 12;
                        const unsigned w0 = (y & 0xFFFFFFu) * DROP_C2 + k2, w1 = (y & 0xFFFFFFu) * DROP_C3 + k3;
                        p[4 * q + 2 * pr] = w0 >= thr32 ? p[4 * q + 2 * pr] : 0.f;
                        p[4 * q + 2 * pr + 1] = w1 >= thr32 ? p[4 * q + 2 * pr + 1] : 0.f;
                    }
            }
            bf16x8 pf[2];
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[ks2][e] = f2bf(p[8 * ks2 + e]);
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x4 lo = tr8(vak[dt][0] + 2048 * ks2 + 4096 * u), hi = tr8(vak[dt][1] + 2048 * ks2 + 4096 * u);
                    bf16x8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3]; vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    O[dt] = mfma32(vf, pf[ks2], O[dt]);
                }
        };

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (active) {                                             // distances of the first sub-tile: two blocks
            band_block(s + 2, 0);
            band_block(s + 1, 1);
        }
        for (int t = 0; t < NT; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of tile t has landed
            __builtin_amdgcn_s_barrier();                         // ... everybody's has; tile t-1 is no longer read
            if (t + 1 < NT) {
                stage_kv(jt_lo + t + 1, (t + 1) & 1);
                stage_rd(-2 - t);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = 2 * t + u;
                if (n < nsub_w) {
                    subtile(t, u);
                    if (n + 1 < nsub_w) band_block(s - u - 2 * t, u);      // the 32 distances the next sub-tile adds
                }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fak[ks] ^= TILEB;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int X = 0; X < 2; ++X) vak[dt][X] ^= TILEB;
        }
        // epilogue: normalise, O^T through the wave's ring area as [32 rows][64] bf16 (chunk c of row r at c ^ (r & 7)),
        // out as whole 128-byte rows; lse
        if (active) {
            const float l = xhalf_sum(lsum);
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
    if (a.drop_thr != 0u) COMMU_LAUNCH((relattn_fwd3_kernel<true>), grid, dim3(512), 0, stream, a);
    else COMMU_LAUNCH((relattn_fwd3_kernel<false>), grid, dim3(512), 0, stream, a);
    COMMU_LAUNCH_CHECK();
    return 0;
}
