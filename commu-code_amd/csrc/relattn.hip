// K6: relative-position masked attention with XL memory (Transformer-XL), flash-style, gfx950.
//
// Reference math (commu/model/model.py:313-345 with _rel_shift :251-259 and the mask of
// :549-574), for one (batch b, head n):
//   S[i,j] = ((q_i+u).k_j + (q_i+v).Rd[i+M-j]) * scale     for j <= i+M,   else masked
//   Rd[d]  = r_net(sinusoid(pos = d))  -- the reference's r[j+T-1-i] re-indexed by DISTANCE
//   P = softmax_j(S),  O_i = sum_j P_ij v_j
// Nothing of shape [B,H,T,K] is materialised in the forward.  The rel-shift is done in
// registers: per 16-row wave tile the (q+v).Rd band product [16 x 80] is computed by MFMA and
// the skewed diagonal  BD[row][jj] = QR[row][row - jj + 63]  is gathered with ds_bpermute (the
// source lane differs only in its low 4 bits) + a select between two adjacent 16-wide blocks.
//
// Layout: activations are time-major rows m = t*B + b (as the reference), so row j of a
// (b,h) matrix is `base + (j*B + b)*ld + h*DH`.  All LDS tiles are row-major copies of global
// tiles ([row][feature], XOR-swizzled 16-byte chunks); an MFMA operand whose contraction index
// is the tile's ROW index (V in P.V, K in dS.K, dO / Q in the dV / dK products) is fetched with
// ds_read_b64_tr_b16 transpose reads, so no transposed copies exist in HBM.
//
// Scores live in the log2 domain: q+u and q+v are pre-multiplied by scale*log2(e) when their MFMA
// fragments are built (forward writes them out as qu2 / qv2 for the backward kernels).
//
// Kernels: relattn_fwd (q-stationary), relattn_bwd_q (q-stationary: the AC part of dq, its column
// sums, and dS indexed by distance for the dR / dq_BD GEMMs), relattn_bwd_kv (kv-stationary: dk,
// dv), attn_delta, transpose_heads.
#include "relattn_common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int PT = 16;      // pitch (elements) of a per-wave transposed tile [64 kv][16 rows]: 4 rows = 128 contiguous bytes per tr-read group

template <int COLS>
__device__ __forceinline__ int swz(int row) {
    constexpr int CH = COLS / 8;
    if (CH >= 16) return row & 15;
    if (CH == 8) return row & 7;
    return ((row >> 3) & 1) * 3;      // 64-byte rows (see gemm.hip swz64)
}
template <int COLS>
__device__ __forceinline__ bf16x8 frag(const bf16* tile, int row, int chunk) {
    constexpr int CH = COLS / 8;
    return ld_bf16x8(tile + row * COLS + ((chunk ^ (swz<COLS>(row) & (CH - 1))) << 3));
}
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x4 tr_read(const bf16* p) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)p));
}
// MFMA operand whose 8 k-slots are ROWS row0..row0+3, row1..row1+3 of a row-major swizzled tile and
// whose 16 rows/cols are the tile COLUMNS col0..col0+15: two transpose reads (lane i of a 16-lane
// group supplies the address of 4 consecutive columns of row (i>>2)).
template <int COLS>
__device__ __forceinline__ bf16x8 frag_tr_rm(const bf16* tile, int row0, int row1, int col0, int r16) {
    constexpr int CH = COLS / 8;
    bf16x8 f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = (h ? row1 : row0) + (r16 >> 2);
        const int col = col0 + 4 * (r16 & 3);
        const bf16x4 v = tr_read(tile + row * COLS + (((col >> 3) ^ (swz<COLS>(row) & (CH - 1))) << 3) + (col & 7));
        f[4 * h + 0] = v[0]; f[4 * h + 1] = v[1]; f[4 * h + 2] = v[2]; f[4 * h + 3] = v[3];
    }
    return f;
}
// Per-wave transposed image T[k][row] (pitch PT = 16 elements = four 8-byte slots per k): slot s of row k lives at
// physical slot s ^ ((k >> 2) & 3).  Without it the 16 lanes of a ds_write_b64 group (k = 16c + r16, same slot) hit
// only 8 of the 32 banks (rows 32 bytes apart: 4-way conflict, a quarter of the forward kernel's LDS cycles).
__device__ __forceinline__ int pt_off(int k, int slot) { return k * PT + ((slot ^ ((k >> 2) & 3)) << 2); }
// A operand (16 rows x 32 k) from a per-wave transposed image T[k][row] (pitch PT)
__device__ __forceinline__ bf16x8 frag_tr(const bf16* img, int kbase, int r16, int g) {
    bf16x8 f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bf16x4 v = tr_read(img + pt_off(kbase + 8 * g + 4 * h + (r16 >> 2), r16 & 3));
        f[4 * h + 0] = v[0]; f[4 * h + 1] = v[1]; f[4 * h + 2] = v[2]; f[4 * h + 3] = v[3];
    }
    return f;
}

// Attention-probability dropout (K16), ONE mask for every attention kernel (both forward generations, backward).
// keep(b, h, i, j): every 32x32 block (i >> 5, j >> 5) of a (batch, head) has a 32-bit key k1 from the strong hash -- block
// coordinates are wave-uniform in all kernels, so that is SCALAR work -- and inside the block a two-round hash on full-rate
// 24-bit multiplies: one mixed word per 2 x 2 cell,
//     y = ((i & 31) >> 1 << 4 | (j & 31) >> 1) * C1 + k1;  y ^= y >> 12;  y &= 0xFFFFFF
// and one multiply-add per element whose constants depend on the element's place in the cell,
//     w = y * CM[i & 1][j & 1] + (k1 * KA[i & 1][j & 1] + KB[i & 1][j & 1]);     keep = w >= round(p * 65536) << 16.
// Whichever two elements of a cell a lane holds -- two ROWS of one key in the 16x16 layout (registers 0,1 / 2,3), two KEYS of
// one query in the transposed 32x32 layout -- share the first round: 2.5 instructions per element in either layout, so the
// forward of one generation and the backward of the other regenerate the same mask.  (Before: one word per row pair in
// the 16x16 family and one per key pair in relattn3.hip -- two masks, and the faster forward unusable in training.)
// Host mirror: ops.attn_dropout_keep_mask.
constexpr unsigned DROP_C1 = 0xD2B74Bu;
constexpr unsigned DROP_CM[2][2] = {{0x9E3779u, 0x85EBCBu}, {0xC2B2AFu, 0xB5297Bu}};
constexpr unsigned DROP_KA[2][2] = {{0x85EBCA6Bu, 0xC2B2AE35u}, {0x27D4EB2Fu, 0x165667B1u}};
constexpr unsigned DROP_KB[2][2] = {{0x6A09E667u, 0xBB67AE85u}, {0x3C6EF372u, 0xA54FF53Au}};
struct DropLane {
    unsigned xc[2];        // ((2 g + rp) << 4 | r16 >> 1) * C1: the lane's two cells of a 16x16 tile (C layout: rows 4g + reg)
    unsigned cm[2];        // CM[row parity][this lane's key parity]
    bool jodd;
    unsigned key_bh;
    __device__ __forceinline__ void init(unsigned seed, int b, int h, int H, int g, int r16) {
        key_bh = mix32(seed + (unsigned)(b * H + h) * 0x9E3779B1u);
        xc[0] = (unsigned)(((2 * g) << 4) | (r16 >> 1)) * DROP_C1;
        xc[1] = (unsigned)(((2 * g + 1) << 4) | (r16 >> 1)) * DROP_C1;
        jodd = (r16 & 1) != 0;
        cm[0] = jodd ? DROP_CM[0][1] : DROP_CM[0][0];
        cm[1] = jodd ? DROP_CM[1][1] : DROP_CM[1][0];
    }
    // hash words of the lane's four elements (rows 4g + reg) of the 16x16 tile (ib, jb) = (i >> 4, j >> 4); both wave-uniform
    __device__ __forceinline__ void words(int ib, int jb, unsigned (&hw)[4]) const {
        const unsigned k1 = mix32k(((unsigned)(ib >> 1) << 16) | (unsigned)(jb >> 1), key_bh);
        // (the tile's place inside its 32x32 block: + 8 row pairs / + 8 key pairs, folded into the additive key)
        const unsigned kk = k1 + (unsigned)((((ib & 1) << 3) << 4) | ((jb & 1) << 3)) * DROP_C1;
        const unsigned ke0 = k1 * DROP_KA[0][0] + DROP_KB[0][0], ko0 = k1 * DROP_KA[0][1] + DROP_KB[0][1];
        const unsigned ke1 = k1 * DROP_KA[1][0] + DROP_KB[1][0], ko1 = k1 * DROP_KA[1][1] + DROP_KB[1][1];
        const unsigned kr0 = jodd ? ko0 : ke0, kr1 = jodd ? ko1 : ke1;
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            unsigned y = xc[rp] + kk;
            y ^= y >> 12;
            y &= 0xFFFFFFu;
            hw[2 * rp] = y * cm[0] + kr0;
            hw[2 * rp + 1] = y * cm[1] + kr1;
        }
    }
};
// reg = row 4g+reg of the tile
__device__ __forceinline__ bool drop_keep16(const unsigned (&hw)[4], int reg, unsigned /*thr*/, unsigned thr_hi) {
    return hw[reg] >= thr_hi;
}

__device__ __forceinline__ float bperm(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// ---------------------------------------------------------------------------------------------
// Tile staging through registers with buffer loads: one SRD per operand, a per-thread byte offset
// computed once, one v_add per load per tile; rows outside the tensor (kv rows >= K, band distances
// d < 0 or d >= K) are outside the SRD's range and read as zero in hardware -- no clamps, no
// branches.  Loads for tile t+1 are issued before tile t is computed and committed to LDS after it.
template <int ROWS, int COLS, int NTHR = 256>
struct Stager {
    static constexpr int CH = COLS / 8;
    static constexpr int N = (ROWS * CH + NTHR - 1) / NTHR;
    static constexpr bool FULL = (ROWS * CH) % NTHR == 0;      // every thread owns N chunks
    static constexpr int RSTEP = NTHR / CH;                    // tile rows between a thread's consecutive chunks
    static_assert(NTHR % CH == 0 && (RSTEP % 16 == 0 || N == 1), "chunk n+1 sits RSTEP rows below chunk n, same swizzle");
    bf16x8 reg[N];
    // One global and one LDS byte offset per thread (chunk 0); chunk n is RSTEP rows further down in both images and
    // -- RSTEP being a multiple of the swizzle period -- keeps the same swizzled column: no per-chunk offset registers.
    unsigned goff0, loff0, gstep;
    bool own;              // (partial tiles only) this thread has a chunk in the last pass
    __device__ __forceinline__ void init(unsigned row_stride_bytes, int tid) {
        const int r = tid / CH, c = tid % CH;
        goff0 = (unsigned)r * row_stride_bytes + (unsigned)c * 16u;
        loff0 = (unsigned)(r * COLS + ((c ^ (swz<COLS>(r) & (CH - 1))) << 3)) * 2u;
        gstep = (unsigned)RSTEP * row_stride_bytes;
        own = FULL || (tid + NTHR * (N - 1) < ROWS * CH);
    }
    __device__ __forceinline__ void load(srd_t srd, unsigned tile_off) {
#pragma unroll
        for (int n = 0; n < N; ++n)      // a thread without a chunk reads far out of range (returns zero, no branch)
            reg[n] = buf_ld(srd, (FULL || n < N - 1 || own) ? goff0 + (unsigned)n * gstep + tile_off : 0xFFFFFFF0u);
    }
    __device__ __forceinline__ void store(bf16* dst) const {
#pragma unroll
        for (int n = 0; n < N; ++n)
            if (FULL || n < N - 1 || own) *(bf16x8*)((char*)dst + loff0 + (unsigned)(n * RSTEP * COLS * 2)) = reg[n];
    }
};

// band row x (0..127) of tile number t lives in a ring of two 64-row halves: half (x>>6) of tile t
// sits at physical half ((x>>6) + t) & 1, so the half shared by consecutive tiles is never moved.
// NCH chunks of 64 rows; STEP = +1 when the band moves up by 64 per tile (kv-stationary loop over query
// tiles), NCH-1 (= -1 mod NCH) when it moves down (q-stationary loops over key tiles).
template <int NCH, int STEP>
__device__ __forceinline__ int ring_row(int x, int t) { return ((((x >> 6) + STEP * t) % NCH) << 6) + (x & 63); }


// =============================================================================================
template <int DH, int NW, bool DROP>
__global__ __launch_bounds__(64 * NW) void relattn_fwd_kernel(const AttnArgs a) {
    constexpr int KS = DH / 32, DB = DH / 16;
    constexpr int QROWS = 16 * NW, NTHR = 64 * NW, NCH = NW / 4 + 1;      // query rows per workgroup, band chunks
    __shared__ __attribute__((aligned(16))) bf16 sK[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sV[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sR[NCH * 64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sP[NW * 64 * PT];

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
    const int QT = (a.T + QROWS - 1) / QROWS;
    // A workgroup takes the query tiles q and QT-1-q of its (batch, head) pair back to back: every workgroup walks the
    // same number of key tiles (the causal triangle folded in half), and half as many workgroups are launched.
    const int QH = (QT + 1) / 2;
    int qslot, h, b;
    tile_coords(QH, a.H, a.B, qslot, h, b);
    for (int rep = 0; rep < 2; ++rep) {
    const int qt = rep == 0 ? QT - 1 - qslot : qslot;
    if (rep == 1 && qt >= QT - 1 - qslot) break;          // odd tile count: the middle tile is done once
    const int i0 = qt * QROWS, T = a.T, M = a.M, B = a.B, K = T + M;
    const bool rst = a.reset != nullptr && a.reset[b] != 0;
    DropLane dl_;
    if (DROP) dl_.init(salted(a.drop_seed), b, h, a.H, g, r16);
    const unsigned thr_hi = a.drop_thr << 16;
    const unsigned rsb = (unsigned)B * a.ld_qkv * 2u;           // bytes between consecutive kv rows
    const float c2 = a.scale * LOG2E;

    bf16x8 qu[KS], qv[KS];
    {
        const int irow = i0 + 16 * w + r16;
        const int iq = min(irow, T - 1);
        const bf16* qp = a.q + ((size_t)iq * B + b) * a.ld_qkv + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 raw = ld_bf16x8(qp + 32 * ks + 8 * g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int f = h * DH + 32 * ks + 8 * g + e;
                const float x = bf2f(raw[e]);
                qu[ks][e] = f2bf((x + a.u[f]) * c2);
                qv[ks][e] = f2bf((x + a.vb[f]) * c2);
            }
            if (a.qu2 != nullptr && irow < T) {
                const size_t off = ((size_t)irow * B + b) * (a.H * DH) + h * DH + 32 * ks + 8 * g;
                st_bf16x8(a.qu2 + off, qu[ks]);
                st_bf16x8(a.qv2 + off, qv[ks]);
            }
        }
    }
    int srcaddr[4];
    bool lower[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        srcaddr[reg] = ((lane & 48) | ((4 * g + reg - 1 - r16) & 15)) << 2;
        lower[reg] = r16 < 4 * g + reg;
    }

    f32x4 o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrow[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    float lpart[4] = {0.f, 0.f, 0.f, 0.f};

    int jt_lo, jt_hi;
    kv_range(a, i0, QROWS, rst, jt_lo, jt_hi);
    bf16* myP = sP + w * 64 * PT;

    const size_t kvbytes = ((size_t)(K - 1) * B * a.ld_qkv + DH) * 2;
    const srd_t srdK = make_srd(a.k + (size_t)b * a.ld_qkv + h * DH, kvbytes);
    const srd_t srdV = make_srd(a.v + (size_t)b * a.ld_qkv + h * DH, kvbytes);
    const srd_t srdR = make_srd(a.rd + h * DH, ((size_t)(K - 1) * a.ld_rd + DH) * 2);
    Stager<64, DH, NTHR> stK, stV, stR;
    stK.init(rsb, tid);
    stV.init(rsb, tid);
    stR.init((unsigned)a.ld_rd * 2u, tid);
    const unsigned rdb = (unsigned)a.ld_rd * 2u;
    auto issue = [&](int jt) {
        const int j0 = jt * 64, dlo = i0 + M - j0 - 63;
        stK.load(srdK, (unsigned)j0 * rsb);
        stV.load(srdV, (unsigned)j0 * rsb);
        stR.load(srdR, (unsigned)dlo * rdb);            // negative dlo wraps: out of range -> zeros
    };
    {   // prologue: high half of the first band, then the first tile
#pragma unroll
        for (int kc = 1; kc < NCH; ++kc) {      // upper chunks of the first band (chunk kc sits in slot kc at t = 0)
            const int dk = i0 + M - jt_lo * 64 - 63 + 64 * kc;
            stR.load(srdR, (unsigned)dk * rdb);
            stR.store(sR + kc * 64 * DH);
        }
        issue(jt_lo);
        stK.store(sK);
        stV.store(sV);
        stR.store(sR);
    }
    __syncthreads();
    const int iw_lo = i0 + 16 * w, iw_hi = iw_lo + 15;
    for (int jt = jt_lo, t = 0; jt <= jt_hi; ++jt, ++t) {
        const int j0 = jt * 64;
        if (jt < jt_hi) issue(jt + 1);

        // band product first; its skewed diagonal BD[row][jj] = QR[row][row - jj + 63] becomes the INITIAL value of
        // the score accumulators, and the QK^T MFMAs accumulate on top (no separate add, 16 fewer live registers)
        f32x4 qr[5];
#pragma unroll
        for (int blk = 0; blk < 5; ++blk) {
            qr[blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int prow = ring_row<NCH, NCH - 1>(16 * w + 16 * blk, t) + r16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qr[blk] = mfma16(qv[ks], frag<DH>(sR, prow, 4 * ks + g), qr[blk]);
        }
        f32x4 s[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            // named scalars (not an array): `cond ? pm[a] : pm[b]` would be turned into a dynamic index
            // select at the SOURCE lane t (dest lane s < row  <=>  t < row), then one permute per output
            const float t0 = lower[reg] ? qr[4][reg] : qr[3][reg], t1 = lower[reg] ? qr[3][reg] : qr[2][reg],
                        t2 = lower[reg] ? qr[2][reg] : qr[1][reg], t3 = lower[reg] ? qr[1][reg] : qr[0][reg];
            s[0][reg] = bperm(srcaddr[reg], t0);
            s[1][reg] = bperm(srcaddr[reg], t1);
            s[2][reg] = bperm(srcaddr[reg], t2);
            s[3][reg] = bperm(srcaddr[reg], t3);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[c] = mfma16(qu[ks], frag<DH>(sK, 16 * c + r16, 4 * ks + g), s[c]);
        }
        const bool need_mask = (j0 + 63 > iw_lo + M) || (a.same_length && j0 <= iw_hi - a.sshift) ||
                               (rst && j0 < M) || (iw_hi >= T);
        if (need_mask) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = iw_lo + 4 * g + reg;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (is_masked(i, j0 + 16 * c + r16, M, a.same_length, a.sshift, rst)) s[c][reg] = -INFINITY;
            }
        }
        // online softmax (log2 domain); rescale only when some row's running max moved
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
                for (int d = 0; d < DB; ++d) o[d][reg] *= alpha;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x4 pb;
            unsigned hw[4] = {0u, 0u, 0u, 0u};
            if (DROP) dl_.words(iw_lo >> 4, (j0 >> 4) + c, hw);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float p = __builtin_amdgcn_exp2f(s[c][reg] - mrow[reg]);
                lpart[reg] += p;                               // the normaliser is the un-dropped sum
                if (DROP) p = drop_keep16(hw, reg, a.drop_thr, thr_hi) ? p : 0.f;      // 1/(1-p) is applied to O at the end
                pb[reg] = f2bf(p);
            }
            *(bf16x4*)(myP + pt_off(16 * c + r16, g)) = pb;       // P^T[kv][row]: rows 4g..4g+3
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 pf = frag_tr(myP, 32 * ks, r16, g);
#pragma unroll
            for (int d = 0; d < DB; ++d)
                o[d] = mfma16(pf, frag_tr_rm<DH>(sV, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * d, r16), o[d]);
        }
        __syncthreads();
        if (jt < jt_hi) {
            stK.store(sK);
            stV.store(sV);
            stR.store(sR + ((((NCH - 1) * (t + 1)) % NCH) << 6) * DH);
        }
        __syncthreads();
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const float l = row16_sum(lpart[reg]);
        const float inv = (DROP ? a.drop_scale : 1.f) / l;
        const int i = i0 + 16 * w + 4 * g + reg;
        if (i < T) {
            bf16* op = a.out + ((size_t)i * B + b) * a.ld_o + h * DH;
#pragma unroll
            for (int d = 0; d < DB; ++d) op[16 * d + r16] = f2bf(o[d][reg] * inv);
            if (r16 == 0) a.lse[((size_t)b * a.H + h) * T + i] = (mrow[reg] + __log2f(l)) * LN2;
        }
    }
    }
}

// =============================================================================================
// Forward, second generation (d_head 64).  Measured on the first-generation kernel above (rocprofv3 PMC, bench
// shape): the VALU is the busiest pipe (~50 %), the MFMA pipe 12 %, and at 192 VGPRs only two waves share a SIMD, so
// nothing hides anything.  This kernel is built around that:
//  * 8 waves x 16 rows = 128 query rows per workgroup, <= 128 VGPRs, two workgroups per CU: four waves per SIMD;
//  * K / V / band tiles arrive by LDS-DMA (buffer_load ... lds; the XOR swizzle is applied on the SOURCE side, the LDS
//    image is lane-linear) into double buffers (the band: a ring of four 64-distance chunks, one new chunk per tile):
//    no staging registers, no LDS store instructions, ONE barrier per key tile;
//  * the softmax is written for instruction count: max3 + DPP-fused row maxima from asm (no canonicalising v_max, no
//    v_mov_dpp), packed subtract / add, the keep-scale of the dropout applied once to O, block-keyed dropout words;
//  * O leaves through the wave's P buffer as whole 128-byte rows.
__device__ __forceinline__ float vmax3(float x, float y, float z) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
__device__ __forceinline__ float vmax2(float x, float y) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// maxima over the 16 lanes of each DPP row, four values at once: the four chains interleave, which covers the two
// wait states a DPP read needs after the VALU write of its source
__device__ __forceinline__ void row16_max4(float& x0, float& x1, float& x2, float& x3) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
}

// STAG: the two halves of the workgroup (waves 0-3 / 4-7: query rows 0-63 / 64-127) run HALF A TILE apart, separated by
// two barriers per key tile: while one half is in the matrix-heavy first half of a tile (band product, skew, QK^T) the
// other is in the VALU-heavy second half (softmax, P image, P.V) of the previous one -- on every SIMD one wave of each.
// K and the band chunk of tile t+1 are requested after the even barrier, V after the odd one (V of tile t-1 is still
// being read by the late half until then), each waited for with a counted vmcnt two barriers later.
template <bool DROP, bool STAG, int WPE = 4>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void relattn_fwd2_kernel(const AttnArgs a) {
    constexpr int DH = 64, NW = 8, QROWS = 128, TILE = 64 * DH, KS = 2, DB = 4;
    __shared__ __attribute__((aligned(1024))) bf16 smem[8 * TILE + NW * 64 * PT];          // 64 + 16 KB
    bf16* const sK = smem;                     // [2][64 keys][64]
    bf16* const sV = smem + 2 * TILE;          // [2][64 keys][64]
    bf16* const sR = smem + 4 * TILE;          // [4][64 distances][64]  ring of band chunks
    bf16* const sP = smem + 8 * TILE;          // per wave: P^T image [64 keys][16 rows], then the O tile [16 rows][64]
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
    const int T = a.T, M = a.M, B = a.B, K = T + M;
    const int QT = (T + QROWS - 1) / QROWS, QH = (QT + 1) / 2;
    int qslot, h, b;
    tile_coords(QH, a.H, B, qslot, h, b);
    const bool rst = a.reset != nullptr && a.reset[b] != 0;
    DropLane dl_;
    if (DROP) dl_.init(salted(a.drop_seed), b, h, a.H, g, r16);
    const unsigned thr_hi = a.drop_thr << 16;
    const unsigned rsb = (unsigned)B * a.ld_qkv * 2u, rdb = (unsigned)a.ld_rd * 2u;
    const float c2 = a.scale * LOG2E;
    const size_t kvbytes = ((size_t)(K - 1) * B * a.ld_qkv + DH) * 2;
    const srd_t srdK = make_srd(a.k + (size_t)b * a.ld_qkv + h * DH, kvbytes);
    const srd_t srdV = make_srd(a.v + (size_t)b * a.ld_qkv + h * DH, kvbytes);
    const srd_t srdR = make_srd(a.rd + h * DH, ((size_t)(K - 1) * a.ld_rd + DH) * 2);
    // LDS-DMA: wave w stages rows 8w..8w+7 of every 64-row tile (one 1 KB instruction per tile and operand); lane l
    // lands at byte 16 l of the wave's slice = row 8w + (l >> 3), physical chunk l & 7, so it fetches source chunk
    // (l & 7) ^ (row & 7).  Rows outside the tensor are outside the descriptor: zeros.
    const int drow = 8 * w + (lane >> 3);
    const unsigned dchunk = (unsigned)(((lane & 7) ^ (lane >> 3)) * 16);
    const unsigned ldsK = lds0 + (unsigned)w * 1024u, ldsV = ldsK + 2u * TILE * 2u, ldsR = ldsK + 4u * TILE * 2u;
    // fragment addressing (elements): row part r16 * 64 + swizzled 16-byte chunk; tile rows are multiples of 16 further
    int foff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = r16 * DH + (((4 * ks + g) ^ (r16 & 7)) << 3);
    int srcaddr[4];
    bool lower[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        srcaddr[reg] = ((lane & 48) | ((4 * g + reg - 1 - r16) & 15)) << 2;
        lower[reg] = r16 < 4 * g + reg;
    }
    bf16* const myP = sP + w * 64 * PT;
    // LDS addressing is spelled out as (lane constant) + (wave-uniform base) + (immediate), in bytes: hipcc otherwise
    // hoists one address register per (k-step, half, feature block) of the transpose reads -- two dozen registers that
    // end up in scratch under the 128-register budget.
    const LDS_AS char* const lds = (const LDS_AS char*)smem;
    //  V^T operand (ds_read_b64_tr_b16 from the row-major tile): rows 32ks + 8g + 4half + (r16 >> 2), features
    //  16d + 4(r16 & 3).  Row, swizzled 16-byte chunk and byte-in-chunk occupy disjoint bit fields of the address and the
    //  feature block d only enters the chunk as (2d) ^ ..., so address(half, d) = vo[half] ^ (32 d): two constants;
    //  k-step ks is +4096 bytes.
    int vo[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int row = 8 * g + 4 * hf + (r16 >> 2), col = 4 * (r16 & 3);
        vo[hf] = row * 128 + (((col >> 3) ^ (row & 7)) << 4) + (col & 7) * 2;
    }
    //  P^T image of the wave (pitch PT, slot swizzle (k >> 2) & 3 = (2g + half) & 3): k-step ks is +1024 bytes
    int po[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) po[hf] = pt_off(8 * g + 4 * hf + (r16 >> 2), r16 & 3) * 2;
    const int pbase = (8 * TILE + w * 64 * PT) * 2;          // the wave's P buffer
    auto tr8 = [&](int byte_off) {
        return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(lds + byte_off)));
    };

    for (int rep = 0; rep < 2; ++rep) {
    const int qt = rep == 0 ? QT - 1 - qslot : qslot;
    if (rep == 1 && qt >= QT - 1 - qslot) break;          // odd tile count: the middle tile is done once
    const int i0 = qt * QROWS;
    const int iw_lo = i0 + 16 * w, iw_hi = iw_lo + 15;
    int jt_lo, jt_hi;
    kv_range(a, i0, QROWS, rst, jt_lo, jt_hi);
    if (rep == 1) __syncthreads();                        // the first tile's buffers are free
    auto stage_kr = [&](int jt, int tt, int chunk) {          // K tile jt and band chunk `chunk` of that tile
        const int j0 = jt * 64, dlo = i0 + M - j0 - 63;
        const unsigned kvoff = (unsigned)(j0 + drow) * rsb + dchunk;
        lds_dma16(srdK, kvoff, ldsK + (unsigned)(tt & 1) * (TILE * 2u));
        lds_dma16(srdR, (unsigned)(dlo + 64 * chunk + drow) * rdb + dchunk, ldsR + (unsigned)((chunk - tt) & 3) * (TILE * 2u));
    };
    auto stage_v = [&](int jt, int tt) {
        lds_dma16(srdV, (unsigned)(jt * 64 + drow) * rsb + dchunk, ldsV + (unsigned)(tt & 1) * (TILE * 2u));
    };
    auto stage = [&](int jt, int tt, int chunk) {
        stage_kr(jt, tt, chunk);
        stage_v(jt, tt);
    };
    {   // prologue: upper band chunks of the first tile, then the tile itself
        const int dlo = i0 + M - jt_lo * 64 - 63;
        lds_dma16(srdR, (unsigned)(dlo + 64 + drow) * rdb + dchunk, ldsR + 1u * (TILE * 2u));
        lds_dma16(srdR, (unsigned)(dlo + 128 + drow) * rdb + dchunk, ldsR + 2u * (TILE * 2u));
        stage(jt_lo, 0, 0);
    }

    bf16x8 qu[KS], qv[KS];
    {
        const int irow = iw_lo + r16;
        const int iq = min(irow, T - 1);
        const bf16* qp = a.q + ((size_t)iq * B + b) * a.ld_qkv + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 raw = ld_bf16x8(qp + 32 * ks + 8 * g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int f = h * DH + 32 * ks + 8 * g + e;
                const float x = bf2f(raw[e]);
                qu[ks][e] = f2bf((x + a.u[f]) * c2);
                qv[ks][e] = f2bf((x + a.vb[f]) * c2);
            }
            if (a.qu2 != nullptr && irow < T) {
                const size_t off = ((size_t)irow * B + b) * (a.H * DH) + h * DH + 32 * ks + 8 * g;
                st_bf16x8(a.qu2 + off, qu[ks]);
                st_bf16x8(a.qv2 + off, qv[ks]);
            }
        }
    }
    f32x4 o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrow[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    f32x2 lp01 = {0.f, 0.f}, lp23 = {0.f, 0.f};

    f32x4 s[4];
    // a wave whose 16 rows see nothing of a key tile only takes part in the staging
    auto sees = [&](int jt) {
        const int j0 = jt * 64;
        return !(j0 > iw_hi + M || (a.same_length && j0 + 63 <= iw_lo - a.sshift) || iw_lo >= T);
    };
    // first half of a tile: band product, skew, QK^T, masks -> s
    auto half1 = [&](int jt, int t) {
        const int j0 = jt * 64;
        const int kb = (t & 1) * (TILE * 2);          // byte base of this tile's K buffer

        // band product, one 16-distance block at a time from the top: block pair (4-c, 3-c) gives the skewed diagonal of
        // score block c, BD[row][jj] = QR[row][row - jj + 63] -- the INITIAL value of its accumulator -- so only two band
        // blocks are live at any time
        auto band = [&](int blk) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            const int x = 16 * w + 16 * blk;                  // band row of the block; chunk x >> 6 sits in slot (chunk - t) & 3
            const int rb = (4 * TILE + ((((x >> 6) - t) & 3) * 64 + (x & 63)) * DH) * 2;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) r = mfma16(qv[ks], *(const LDS_AS bf16x8*)(lds + rb + 2 * foff[ks]), r);
            return r;
        };
        auto skew = [&](const f32x4& hi, const f32x4& lo) {          // select at the SOURCE lane, then one permute per output
            f32x4 r;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) r[reg] = bperm(srcaddr[reg], lower[reg] ? hi[reg] : lo[reg]);
            return r;
        };
        {
            f32x4 qa = band(4), qb = band(3);
            s[0] = skew(qa, qb);
            qa = band(2);
            s[1] = skew(qb, qa);
            qb = band(1);
            s[2] = skew(qa, qb);
            qa = band(0);
            s[3] = skew(qb, qa);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[c] = mfma16(qu[ks], *(const LDS_AS bf16x8*)(lds + kb + 2 * foff[ks] + 2048 * c), s[c]);
        }
        const bool need_mask = (j0 + 63 > iw_lo + M) || (a.same_length && j0 <= iw_hi - a.sshift) ||
                               (rst && j0 < M) || (iw_hi >= T);
        if (need_mask) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = iw_lo + 4 * g + reg;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (is_masked(i, j0 + 16 * c + r16, M, a.same_length, a.sshift, rst)) s[c][reg] = -INFINITY;
            }
        }
    };
    // second half: online softmax, P image, P.V
    auto half2 = [&](int jt, int t) {
        const int j0 = jt * 64;
        const int vb = 2 * TILE * 2 + (t & 1) * (TILE * 2);          // byte base of this tile's V buffer
        // online softmax (log2 domain); rescale only when some row's running max moved
        float mx[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) mx[reg] = vmax2(vmax3(s[0][reg], s[1][reg], s[2][reg]), s[3][reg]);
        row16_max4(mx[0], mx[1], mx[2], mx[3]);
        bool grew = false;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            mx[reg] = vmax2(mrow[reg], mx[reg]);
            grew |= mx[reg] > mrow[reg];
        }
        if (__any(grew)) {
            float al[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                al[reg] = __builtin_amdgcn_exp2f(mrow[reg] - mx[reg]);
                mrow[reg] = mx[reg];
#pragma unroll
                for (int d = 0; d < DB; ++d) o[d][reg] *= al[reg];
            }
            lp01 *= (f32x2){al[0], al[1]};
            lp23 *= (f32x2){al[2], al[3]};
        }
        const f32x2 m01 = {mrow[0], mrow[1]}, m23 = {mrow[2], mrow[3]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned hw[4] = {0u, 0u, 0u, 0u};
            if (DROP) dl_.words(iw_lo >> 4, (j0 >> 4) + c, hw);
            const f32x2 e01 = (f32x2){s[c][0], s[c][1]} - m01, e23 = (f32x2){s[c][2], s[c][3]} - m23;
            f32x2 p01 = {__builtin_amdgcn_exp2f(e01[0]), __builtin_amdgcn_exp2f(e01[1])};
            f32x2 p23 = {__builtin_amdgcn_exp2f(e23[0]), __builtin_amdgcn_exp2f(e23[1])};
            lp01 += p01;                                        // the normaliser is the un-dropped sum
            lp23 += p23;
            if (DROP) {                                         // 1/(1-p) is applied to O at the end
                p01[0] = drop_keep16(hw, 0, a.drop_thr, thr_hi) ? p01[0] : 0.f;
                p01[1] = drop_keep16(hw, 1, a.drop_thr, thr_hi) ? p01[1] : 0.f;
                p23[0] = drop_keep16(hw, 2, a.drop_thr, thr_hi) ? p23[0] : 0.f;
                p23[1] = drop_keep16(hw, 3, a.drop_thr, thr_hi) ? p23[1] : 0.f;
            }
            bf16x4 pb;
            pb[0] = f2bf(p01[0]); pb[1] = f2bf(p01[1]); pb[2] = f2bf(p23[0]); pb[3] = f2bf(p23[1]);
            *(bf16x4*)(myP + pt_off(16 * c + r16, g)) = pb;       // P^T[kv][row]: rows 4g..4g+3
        }
        __builtin_amdgcn_wave_barrier();
        // (requesting ALL operand fragments of a phase ahead of its MFMAs -- source-level batching plus
        //  sched_group_barrier -- was measured: the extra live registers cost more than the exposed LDS latency at four
        //  waves per SIMD; two feature blocks at a time fits the register budget and is worth ~2 %)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 pf;
            {
                const bf16x4 lo = tr8(pbase + po[0] + 1024 * ks), hi = tr8(pbase + po[1] + 1024 * ks);
                pf[0] = lo[0]; pf[1] = lo[1]; pf[2] = lo[2]; pf[3] = lo[3]; pf[4] = hi[0]; pf[5] = hi[1]; pf[6] = hi[2]; pf[7] = hi[3];
            }
#pragma unroll
            for (int d2 = 0; d2 < DB; d2 += 2) {          // two feature blocks' V fragments (4 transpose reads) ahead of their MFMAs
                bf16x8 vf[2];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const bf16x4 lo = tr8(((vb + vo[0]) ^ (32 * (d2 + d))) + 4096 * ks), hi = tr8(((vb + vo[1]) ^ (32 * (d2 + d))) + 4096 * ks);
                    vf[d][0] = lo[0]; vf[d][1] = lo[1]; vf[d][2] = lo[2]; vf[d][3] = lo[3];
                    vf[d][4] = hi[0]; vf[d][5] = hi[1]; vf[d][6] = hi[2]; vf[d][7] = hi[3];
                }
#pragma unroll
                for (int d = 0; d < 2; ++d) o[d2 + d] = mfma16(pf, vf[d], o[d2 + d]);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();                         // P image is rewritten by the next tile
    };
    if (!STAG) {
        for (int jt = jt_lo, t = 0; jt <= jt_hi; ++jt, ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share of tile t has landed
            __builtin_amdgcn_s_barrier();                            // ... everybody's has; tile t-1 is no longer read
            if (jt < jt_hi) stage(jt + 1, t + 1, 0);
            if (!sees(jt)) continue;
            half1(jt, t);
            half2(jt, t);
        }
    } else {
        // barrier 2t: K / band of tile t (and, for t = 0, V) have landed; barrier 2t+1: V of tile t has landed.
        // early half (waves 0-3): half1(t) after barrier 2t, half2(t) after 2t+1; late half: half1(t) after 2t+1,
        // half2(t) after 2t+2.  NT tiles -> barriers 0 .. 2 NT.
        const int NT = jt_hi - jt_lo + 1;
        const int late = w >= 4 ? 1 : 0;
        for (int n = 0; n <= 2 * NT; ++n) {
            const int te = n >> 1;                                   // tile whose barrier pair this is
            if (!(n & 1)) {
                // even barrier: outstanding = [K, R of tile te] (+ V of tile te, requested after them, when te >= 1)
                if (te >= 1 && te < NT) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (te + 1 < NT) stage_kr(jt_lo + te + 1, te + 1, 0);      // K buffer / band slot of tile te-1: last read before this barrier
            } else {
                // odd barrier: V of tile te must have landed; K, R of tile te+1 (2 requests) may still be in flight
                if (te + 1 < NT) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (te + 1 < NT) stage_v(jt_lo + te + 1, te + 1);          // V buffer of tile te-1: the late half finished it before this barrier
            }
            const int m = n - late;
            if (m < 0 || m >= 2 * NT) continue;
            const int t = m >> 1;
            if (!sees(jt_lo + t)) continue;
            if (m & 1) half2(jt_lo + t, t);
            else half1(jt_lo + t, t);
        }
    }
    // epilogue: normalise, O through the wave's P buffer ([16 rows][64] bf16, 16-byte chunk c of row r at chunk
    // c ^ (r & 7)), out as whole 128-byte rows; lse
    {
        float lr[4] = {lp01[0], lp01[1], lp23[0], lp23[1]};
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float l = row16_sum(lr[reg]);
            const float inv = (DROP ? a.drop_scale : 1.f) / l;
            const int row = 4 * g + reg, i = iw_lo + row;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const int col = 16 * d + r16;
                myP[row * DH + ((((col >> 3) ^ (row & 7))) << 3) + (col & 7)] = f2bf(o[d][reg] * inv);
            }
            if (r16 == 0 && i < T) a.lse[((size_t)b * a.H + h) * T + i] = (mrow[reg] + __log2f(l)) * LN2;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int row = (lane >> 3) + 8 * n, ch = lane & 7, i = iw_lo + row;
            if (i < T)
                st_bf16x8(a.out + ((size_t)i * B + b) * a.ld_o + h * DH + 8 * ch, ld_bf16x8(myP + row * DH + ((ch ^ (row & 7)) << 3)));
        }
        __builtin_amdgcn_wave_barrier();
    }
    }
}

// =============================================================================================
// backward, q-stationary: dq_AC = dS.K (+ its column sums for d r_w_bias) and dS written by
// DISTANCE (dSk[i][d = i+M-j]) for the two GEMMs  dq_BD = dSk.Rd  and  dRd = dSk^T.(q+v).
// FROMP (d_head 64, four waves, a.pf from commu_relattn_fwd_save): the probabilities come from the FORWARD pass -- per key
// tile each wave brings ONE 2176-byte tile of a.pf (32 queries x 32 keys in the forward kernel's accumulator order + the 32
// running maxima) to LDS by LDS-DMA, double buffered, and every lane gathers its 16 values of the 16 x 64 block with
// 2-byte LDS reads at immediate offsets from one lane address: no (q + u) . k product, no band product, no rel-shift
// permutes, no exponential per element, no dropout hash (8 of the 26 score MFMAs and no Rd staging remain).
template <int DH, int NW, bool DROP, bool FROMP = false>
// (waves_per_eu(2, 2): the kernel sits at the 256-register edge; without the bound a small edit lets the compiler take a
//  few registers more and the workgroup silently drops to one wave per SIMD -- measured 1.53 -> 1.73 ms per layer pass)
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void relattn_bwd_q_kernel(const AttnArgs a) {
    static_assert(!FROMP || (DH == 64 && NW == 4), "forward-saved probabilities: d_head 64, 64 query rows per workgroup");
    constexpr int KS = DH / 32, DB = DH / 16;
    constexpr int QROWS = 16 * NW, NTHR = 64 * NW, NCH = NW / 4 + 1;      // query rows per workgroup, band chunks
    constexpr int PFBUF = 4 * PF_TILE_BYTES;                              // FROMP: (2 blocks of 32 queries) x (2 sub-tiles of 32 keys)
    __shared__ __attribute__((aligned(16))) bf16 sK[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sV[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sR[FROMP ? 8 : NCH * 64 * DH];
    __shared__ __attribute__((aligned(16))) char sPF[FROMP ? 2 * PFBUF : 16];
    __shared__ __attribute__((aligned(16))) bf16 sD[NW * 64 * PT];
    __shared__ float red[NW][DH];
    // dS-by-distance leaves through a per-wave ring [16 rows][128 distances] (row r holds distance d at column
    // (d + 32 (r >> 2)) mod 128: the four row groups of a write land 16 banks apart, rows are 256 bytes so the column bits
    // and the row bits of the byte address do not overlap).  Element (row, jj) of a key tile IS distance
    // d = i + M - j0 - jj: it is written straight to its column -- the skew is absorbed by the LDS address: one add and one
    // and-or per element, no permutes, no selects, no range checks; masked positions carry dS = 0 -- and leaves for HBM
    // as whole aligned 16-byte chunks, 8 per row and key tile (the chunks the tile completed).
    constexpr int SRING = 128, SPITCH = 128;
    __shared__ __attribute__((aligned(256))) bf16 sS[NW * 16 * SPITCH];

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
    const int QT = (a.T + QROWS - 1) / QROWS;
    const int QH = (QT + 1) / 2;          // query tiles q and QT-1-q back to back (see relattn_fwd_kernel)
    int qslot, h, b;
    tile_coords(QH, a.H, a.B, qslot, h, b);
    bf16* myS = sS + w * 16 * SPITCH;
    const unsigned sbase = (unsigned)(size_t)(LDS_AS bf16*)sS + (unsigned)(w * 16 * SPITCH * 2 + 1024 * (int)((threadIdx.x & 63) >> 4));
    for (int rep = 0; rep < 2; ++rep) {
    const int qt = rep == 0 ? QT - 1 - qslot : qslot;
    if (rep == 1 && qt >= QT - 1 - qslot) break;
    const int i0 = qt * QROWS, T = a.T, M = a.M, B = a.B, K = T + M;
    const bool rst = a.reset != nullptr && a.reset[b] != 0;
    DropLane dl_;
    if (DROP && !FROMP) dl_.init(salted(a.drop_seed), b, h, a.H, g, r16);
    const unsigned thr_hi = a.drop_thr << 16;
    const unsigned rsb = (unsigned)B * a.ld_qkv * 2u;
    const int HD = a.H * DH;
    // FROMP: this wave's tile stream -- block (i0 >> 5) + (w >> 1) of the pair, sub-tile 2 jt + (w & 1) of every key tile;
    // a block row beyond T (or a sub-tile beyond K) lies outside the descriptor and reads as zero
    const int NS32 = (K + 31) >> 5, NB32 = (T + 31) >> 5;
    const int pblk = (i0 >> 5) + (w >> 1);
    const srd_t srdPF = make_srd((const char*)a.pf + (((size_t)b * a.H + h) * NB32 + (size_t)min(pblk, NB32 - 1)) * (size_t)NS32 * PF_TILE_BYTES,
                                 (FROMP && pblk < NB32) ? (size_t)NS32 * PF_TILE_BYTES : 0);
    const unsigned pf_lds = (unsigned)(size_t)(LDS_AS char*)sPF + (unsigned)(w * PF_TILE_BYTES);
    // gather address of this lane (see the kernel comment): element (row 4g + reg, key 16 c + r16) of the wave's block
    // sits at pfl + reg * 32 + (c >> 1) * PF_TILE_BYTES + (c & 1) * 16 of the tile buffer
    const int pfl = (w >> 1) * 2 * PF_TILE_BYTES + (16 * (w & 1) + 4 * g) * 32 + 1024 * ((r16 >> 2) & 1) +
                    (4 * (r16 >> 3) + 3 - (r16 & 3)) * 2;
    const int mfl = (w >> 1) * 2 * PF_TILE_BYTES + 2048 + (16 * (w & 1) + 4 * g) * 4;
    // dS = P (keep dP/(1-p) - delta) scale  =  [P scale/(1-p)] (keep dP - delta (1-p)): the constant factor goes into the
    // exponent (lse2 below), delta is pre-multiplied per row -- per element: exp2, select, subtract, multiply
    const float dsc = DROP ? a.drop_scale : 1.f;

    // every global load of the prologue is requested before anything waits (a workgroup lives ~30 us and its prologue is
    // exposed): the first K / V / band tile and the band's upper chunk(s) here, then the query-side fragments and the
    // row statistics; the LDS ring is cleared and the statistics are folded while they are in flight
    int jt_lo, jt_hi;
    kv_range(a, i0, QROWS, rst, jt_lo, jt_hi);
    const size_t kvbytes = ((size_t)(K - 1) * B * a.ld_qkv + DH) * 2;
    const srd_t srdK = make_srd(a.k + (size_t)b * a.ld_qkv + h * DH, kvbytes);
    const srd_t srdV = make_srd(a.v + (size_t)b * a.ld_qkv + h * DH, kvbytes);
    const srd_t srdR = make_srd(a.rd + h * DH, ((size_t)(K - 1) * a.ld_rd + DH) * 2);
    Stager<64, DH, NTHR> stK, stV, stR, stU[NCH - 1];
    stK.init(rsb, tid);
    stV.init(rsb, tid);
    stR.init((unsigned)a.ld_rd * 2u, tid);
    const unsigned rdb = (unsigned)a.ld_rd * 2u;
    auto issue = [&](int jt, int buf) {
        const int j0 = jt * 64, dlo = i0 + M - j0 - 63;
        if (FROMP) {
            // (BEFORE the register-staged loads: loads return in order, so the counted wait in front of stK.store() -- the
            //  compiler counts its own later memory operations -- also covers these three)
            const unsigned so = (unsigned)(2 * jt + (w & 1)) * (unsigned)PF_TILE_BYTES, dst = pf_lds + (unsigned)(buf * PFBUF);
            lds_dma16s(srdPF, (unsigned)lane * 16u, so, dst);
            lds_dma16s(srdPF, 1024u + (unsigned)lane * 16u, so, dst + 1024u);
            if (lane < 8) lds_dma16s(srdPF, 2048u + (unsigned)lane * 16u, so, dst + 2048u);
        }
        stK.load(srdK, (unsigned)j0 * rsb);
        stV.load(srdV, (unsigned)j0 * rsb);
        if (!FROMP) stR.load(srdR, (unsigned)dlo * rdb);
    };
    issue(jt_lo, 0);
    if (!FROMP) {
#pragma unroll
        for (int kc = 1; kc < NCH; ++kc) {      // upper chunks of the first band
            stU[kc - 1].init((unsigned)a.ld_rd * 2u, tid);
            stU[kc - 1].load(srdR, (unsigned)(i0 + M - jt_lo * 64 - 63 + 64 * kc) * rdb);
        }
    }

    bf16x8 qu[KS], qv[KS], dof[KS];
    float drow = 0.f;          // a.o_in: delta of row r16 (sum over d of o . dout), in every lane of that row after the reduction
    {
        const int iq = min(i0 + 16 * w + r16, T - 1);
        const size_t off = ((size_t)iq * B + b) * HD + h * DH;
        const bf16* dop = a.dout + ((size_t)iq * B + b) * a.ld_o + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (!FROMP) {
                qu[ks] = ld_bf16x8(a.qu2 + off + 32 * ks + 8 * g);
                qv[ks] = ld_bf16x8(a.qv2 + off + 32 * ks + 8 * g);
            }
            dof[ks] = ld_bf16x8(dop + 32 * ks + 8 * g);
        }
        if (a.o_in != nullptr) {
            const bf16* op = a.o_in + ((size_t)iq * B + b) * a.ld_o + h * DH;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 of = ld_bf16x8(op + 32 * ks + 8 * g);
#pragma unroll
                for (int e = 0; e < 8; ++e) drow += bf2f(of[e]) * bf2f(dof[ks][e]);
            }
            drow += __shfl_xor(drow, 16, 64);
            drow += __shfl_xor(drow, 32, 64);
            if (g == 0 && i0 + 16 * w + r16 < T)
                const_cast<float*>(a.delta)[((size_t)b * a.H + h) * T + i0 + 16 * w + r16] = drow;
        }
    }
    // -lse2 and -delta/dsc are the INITIAL values of the score / dP accumulators (the MFMA's C operand): no subtraction per element
    f32x4 nls, ndl;
    int srcaddr[4];
    bool lower[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int i = min(i0 + 16 * w + 4 * g + reg, T - 1);
        nls[reg] = __log2f(a.scale * dsc) - a.lse_in[((size_t)b * a.H + h) * T + i] * LOG2E;
        // (rows 4g + reg of the C layout: from the lane that holds that row)
        ndl[reg] = -(a.o_in != nullptr ? bperm((4 * g + reg) << 2, drow) : a.delta[((size_t)b * a.H + h) * T + i]) / dsc;
        srcaddr[reg] = ((lane & 48) | ((4 * g + reg - 1 - r16) & 15)) << 2;
        lower[reg] = r16 < 4 * g + reg;
    }
    // P scratch for relattn_bwd_kv2_kernel: this (batch, head)'s [ceil(T/16)][ceil(K/64)] blocks of 2 KB
    const int JT = (K + 63) >> 6;
    // The P stores are issued UNCONDITIONALLY (a wave without rows, or a launch without a P buffer, stores through an
    // empty descriptor: dropped by the range check).  A store inside an `if` is not counted by the compiler's vmcnt
    // bookkeeping as certainly younger than the staging loads, so its counted waits for those loads (bottom of the tile)
    // also waited for this tile's four P stores to be ACKNOWLEDGED -- a round trip to memory per tile.
    const bool pstore = a.pbuf != nullptr && i0 + 16 * w < T;
    // (p_layout 1, relattn_bwd_kv3_kernel: 2-KB blocks of 32 queries x 32 keys; this wave's rows are half (w & 1) of block row
    //  iw_lo >> 5, key block c of the tile is half (c & 1) of block column 2 jt + (c >> 1); a lane's 8 bytes -- four queries of
    //  one key -- sit at key * 64 + (g & 1) * 32 + (2 (w & 1) + (g >> 1)) * 8 of the block)
    const size_t pblocks = a.p_layout ? (size_t)(2 * ((T + 31) >> 5)) * JT : (size_t)((T + 15) >> 4) * JT;
    const srd_t srdP = make_srd(a.pbuf + ((size_t)b * a.H + h) * pblocks * 1024, pstore ? pblocks * 2048 : 0);
    const int pvoff = a.p_layout ? r16 * 64 + (g & 1) * 32 + (2 * (w & 1) + (g >> 1)) * 8
                                 : pt_off(r16, g) * 2;          // bytes; pt_off(16c + r16, g) = 256 c + pt_off(r16, g)
    int foff[KS];          // fragment addressing: lane part r16 * DH + swizzled chunk (block rows are multiples of 16)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = r16 * DH + (((4 * ks + g) ^ (swz<DH>(r16) & (DH / 8 - 1))) << 3);
    f32x4 dq[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) dq[d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16* myD = sD + w * 64 * PT;
    for (int n = lane; n < 16 * SPITCH / 8; n += 64) *(bf16x8*)(myS + n * 8) = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    {
        if (!FROMP) {
#pragma unroll
            for (int kc = 1; kc < NCH; ++kc) stU[kc - 1].store(sR + kc * 64 * DH);      // (chunk kc sits in slot kc at t = 0)
        }
        stK.store(sK);
        stV.store(sV);
        if (!FROMP) stR.store(sR);
    }
    __syncthreads();
    const int iw_lo = i0 + 16 * w, iw_hi = iw_lo + 15;
    const size_t mrow0 = (size_t)T * B;
    // dS-by-distance flush: descriptor of this head's [T*B][ld_dsk] (tiled or row-major) block, per-lane row offset in bytes
    const srd_t srdD = make_srd(a.dsk + (size_t)h * mrow0 * a.ld_dsk, mrow0 * a.ld_dsk * 2);
    const int fl_i = iw_lo + (lane >> 2);
    unsigned fl_row;
    {
        const unsigned m = (unsigned)min(fl_i, T - 1) * (unsigned)B + (unsigned)b;
        fl_row = a.dsk_tiled ? (((m >> 6) * (unsigned)(a.ld_dsk >> 7)) << 14) + ((m & 63u) << 8) : m * (unsigned)a.ld_dsk * 2u;
    }
    if (rst && jt_lo > 0) {
        // a sequence that starts here (reset_mems) does not see the memory: its key tiles below jt_lo are skipped -- their
        // distances (dtop, i + M] of every row must still READ as zero for the band consumers, and the scratch is re-used
        // between layers and steps, so they are written: whole 16-byte chunks above the one the first tile completes
        const int dtop = fl_i + M - 64 * jt_lo, clast = (fl_i + M) >> 3;
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (int c = (dtop >> 3) + 1 + (lane & 3); c <= ((iw_hi + M) >> 3); c += 4) {
            unsigned off = a.dsk_tiled ? fl_row + (((unsigned)(c >> 4) << 13) + (unsigned)((8 * c) & 127)) * 2u
                                       : fl_row + (unsigned)(8 * c) * 2u;
            if (!(fl_i < T && c <= clast)) off = 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(z4, srdD, (int)off, 0, 0);
        }
    }
    for (int jt = jt_lo, t = 0; jt <= jt_hi; ++jt, ++t) {
        const int j0 = jt * 64;
        if (jt < jt_hi) issue(jt + 1, (t + 1) & 1);

        f32x4 s[4], dp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s[c] = nls;
            dp[c] = ndl;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (!FROMP) s[c] = mfma16(qu[ks], ld_bf16x8(sK + 16 * c * DH + foff[ks]), s[c]);
                dp[c] = mfma16(dof[ks], ld_bf16x8(sV + 16 * c * DH + foff[ks]), dp[c]);
            }
        }
        const bool need_mask = (j0 + 63 > iw_lo + M) || (a.same_length && j0 <= iw_hi - a.sshift) ||
                               (rst && j0 < M) || (iw_hi >= T);
        if (!FROMP) {
            f32x4 qr[5];
#pragma unroll
            for (int blk = 0; blk < 5; ++blk) {
                qr[blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const bf16* rb = sR + ring_row<NCH, NCH - 1>(16 * w + 16 * blk, t) * DH;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) qr[blk] = mfma16(qv[ks], ld_bf16x8(rb + foff[ks]), qr[blk]);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                // select at the SOURCE lane t (dest lane s < row  <=>  t < row), then one permute per output
                const float t0 = lower[reg] ? qr[4][reg] : qr[3][reg], t1 = lower[reg] ? qr[3][reg] : qr[2][reg],
                            t2 = lower[reg] ? qr[2][reg] : qr[1][reg], t3 = lower[reg] ? qr[1][reg] : qr[0][reg];
                s[0][reg] += bperm(srcaddr[reg], t0);
                s[1][reg] += bperm(srcaddr[reg], t1);
                s[2][reg] += bperm(srcaddr[reg], t2);
                s[3][reg] += bperm(srcaddr[reg], t3);
            }
            if (need_mask) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int i = iw_lo + 4 * g + reg;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (is_masked(i, j0 + 16 * c + r16, M, a.same_length, a.sshift, rst) || i >= T) s[c][reg] = -INFINITY;
                }
            }
        }
        // FROMP: P scale / (1-p) = |e| * exp2(m + nls) with e, m from the forward pass; eight factors per lane and tile
        const LDS_AS char* pfb = (const LDS_AS char*)sPF + (t & 1) * PFBUF;
        f32x4 fac[2];
        if (FROMP) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const f32x4 mv = *(const LDS_AS f32x4*)(pfb + mfl + sub * PF_TILE_BYTES);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) fac[sub][reg] = __builtin_amdgcn_exp2f(mv[reg] + nls[reg]);
            }
        }
        // dS = P (dP - delta) scale
        // byte column of (row 4g, jj = r16) in the rotated ring, before the wrap: 2 (distance + 32 g); sbase: the LDS byte
        // address of row 4g (a multiple of 256: the column byte is OR-ed in)
        const int dcolb = 2 * (iw_lo + 4 * g + M - j0 - r16 + 32 * g);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x4 db;
            unsigned hw[4] = {0u, 0u, 0u, 0u};
            if (DROP && !FROMP) dl_.words(iw_lo >> 4, (j0 >> 4) + c, hw);
            bf16x4 pq;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float p;          // P * scale / (1-p)
                bool keep;
                if (FROMP) {
                    const unsigned raw = *(const LDS_AS unsigned short*)(pfb + pfl + reg * 32 + (c >> 1) * PF_TILE_BYTES + (c & 1) * 16);
                    keep = (raw & 0x8000u) == 0u;
                    p = __builtin_bit_cast(float, (raw & 0x7FFFu) << 16) * fac[c >> 1][reg];
                    // (tiles the forward pass never wrote -- beyond the causal edge of a 32-row block -- hold anything: select)
                    if (need_mask) {
                        const int i = iw_lo + 4 * g + reg;
                        if (is_masked(i, j0 + 16 * c + r16, M, a.same_length, a.sshift, rst) || i >= T) p = 0.f;
                    }
                } else {
                    p = __builtin_amdgcn_exp2f(s[c][reg]);
                    keep = !DROP || drop_keep16(hw, reg, a.drop_thr, thr_hi);
                }
                float dpe = dp[c][reg];                                       // dP - delta (1-p)
                float ps = p;
                if (DROP) {
                    dpe = keep ? dpe : ndl[reg];
                    ps = keep ? p : -p;          // the stored probability carries the keep decision in its sign: the
                }                                // key-stationary kernel does not hash again
                db[reg] = f2bf(p * dpe);
                pq[reg] = f2bf(ps);
            }
            // the key-stationary kernel re-reads P instead of recomputing it (block of 16 rows x 64 keys, P^T image order)
            if (DH == 64)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pq), srdP,
                                                      pvoff + (a.p_layout ? 1024 * (c & 1) + 2048 * (c >> 1) : 512 * c),
                                                      a.p_layout ? (((iw_lo >> 5) * 2 * JT + 2 * jt) << 11) : (((iw_lo >> 4) * JT + jt) << 11),
                                                      2 /* nt: read once, by a later kernel */);
            *(bf16x4*)(myD + pt_off(16 * c + r16, g)) = db;       // dS^T[kv][row]
            // by distance: (row 4g+reg, jj = 16c + r16) -> ring column (i + M - j0 - jj) & 127
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                ((LDS_AS bf16*)(size_t)(((unsigned)(dcolb + 2 * reg - 32 * c) & 0xFEu) | sbase))[128 * reg] = db[reg];
        }
        __builtin_amdgcn_wave_barrier();
        // flush: row r (distance dl = i + M - j0 at jj = 0) completed the aligned chunks 8c in [dl - 63, dl]
        {
            // lane (row = lane >> 2, k = lane & 3): chunks c = c0 + k and c0 + k + 4 of its row; the row's part of the address
            // (fl_row) is per query tile, the descriptor is per head, a false predicate becomes an out-of-range offset
            const int dl0 = fl_i + M - j0;
            const int c0 = ((dl0 - 56) >> 3) + (lane & 3);          // ceil((dl - 63) / 8) + k
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int c = c0 + 4 * n;
                const bf16x8 v8 = *(const bf16x8*)(myS + (lane >> 2) * SPITCH + ((8 * c + 32 * (lane >> 4)) & (SRING - 1)));
                unsigned off = a.dsk_tiled ? fl_row + (((unsigned)(c >> 4) << 13) + (unsigned)((8 * c) & 127)) * 2u
                                           : fl_row + (unsigned)(8 * c) * 2u;
                if (!(fl_i < T && c >= 0 && 8 * c <= dl0)) off = 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v8), srdD, (int)off, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 df = frag_tr(myD, 32 * ks, r16, g);
#pragma unroll
            for (int d = 0; d < DB; ++d)
                dq[d] = mfma16(df, frag_tr_rm<DH>(sK, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * d, r16), dq[d]);
        }
        __syncthreads();
        if (jt < jt_hi) {
            stK.store(sK);
            stV.store(sV);
            if (!FROMP) stR.store(sR + ((((NCH - 1) * (t + 1)) % NCH) << 6) * DH);
        }
        __syncthreads();
    }
    if (a.dsk_wedge > 0) {          // zeros right of the causal edge, as far as the band pass / GEMMs read
        for (int r = 0; r < 16; ++r) {
            const int i = iw_lo + r;
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
    // dq (AC part) leaves through the wave's dS^T scratch as whole 16-byte pieces (sixteen 2-byte stores per lane before)
    static_assert(64 * PT >= 16 * DH, "the per-wave scratch holds a 16 x DH tile");
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 4 * g + reg, col = 16 * d + r16;
            myD[row * DH + ((((col >> 3) ^ (row & 7)) & (DH / 8 - 1)) << 3) + (col & 7)] = f2bf(dq[d][reg]);
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n = 0; n < DH / 32; ++n) {          // 16 rows x DH / 8 pieces over 64 lanes
        const int idx = lane + 64 * n, row = idx / (DH / 8), ch = idx % (DH / 8), i = iw_lo + row;
        if (i < T)
            st_bf16x8(a.dq + ((size_t)i * B + b) * HD + h * DH + 8 * ch,
                      ld_bf16x8(myD + row * DH + (((ch ^ (row & 7)) & (DH / 8 - 1)) << 3)));
    }
#pragma unroll
    for (int d = 0; d < DB; ++d) {
        float ca = 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = i0 + 16 * w + 4 * g + reg;
            if (i < T) ca += dq[d][reg];
        }
        ca += __shfl_xor(ca, 16, 64);
        ca += __shfl_xor(ca, 32, 64);
        if (g == 0) red[w][16 * d + r16] = ca;
    }
    __syncthreads();
    if (tid < DH)
    {
        float acc = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) acc += red[ww][tid];
        a.du_part[((size_t)b * QT + qt) * HD + h * DH + tid] = acc;
    }
    __syncthreads();          // `red` is reused by the second tile
    }
}

// =============================================================================================
// backward, kv-stationary: dk, dv.  Wave w owns kv columns 16w..16w+15 of the tile; S, dP are
// held as [64 q rows x 16 kv cols] in C layout, which IS the A-operand layout of P^T / dS^T.
template <int DH, int NW, bool DROP>
__global__ __launch_bounds__(64 * NW) void relattn_bwd_kv_kernel(const AttnArgs a) {
    constexpr int KS = DH / 32, DB = DH / 16;
    constexpr int KCOLS = 16 * NW, NTHR = 64 * NW, NCH = NW / 4 + 1;      // key columns per workgroup, band chunks
    __shared__ __attribute__((aligned(16))) bf16 sQu[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sQv[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sdO[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sR[NCH * 64 * DH];
    __shared__ __attribute__((aligned(16))) float sLse[64], sDl[64];

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
    const int NT = (a.T + a.M + KCOLS - 1) / KCOLS, NH = (NT + 1) / 2;      // key tiles j and NT-1-j back to back
    int jslot, h, b;
    tile_coords(NH, a.H, a.B, jslot, h, b);
    for (int rep = 0; rep < 2; ++rep) {
    const int jt = rep == 0 ? jslot : NT - 1 - jslot;
    if (rep == 1 && jt <= jslot) break;
    const int j0 = jt * KCOLS, T = a.T, M = a.M, B = a.B, K = T + M;
    const bool rst = a.reset != nullptr && a.reset[b] != 0;
    DropLane dl_;
    if (DROP) dl_.init(salted(a.drop_seed), b, h, a.H, g, r16);
    const unsigned thr_hi = a.drop_thr << 16;
    const int HD = a.H * DH;
    // With P' = P ln2/(1-p) (the factor goes into the exponent through sLse):  dS'' = P' (keep dP - delta (1-p))  and
    // dV = [sum keep P' dO] / ln2 (applied to the accumulator at the end) -- per element: exp2, two selects, subtract,
    // multiply; (q+u) is pre-scaled by scale*log2e, hence the ln2
    const float dsc = DROP ? a.drop_scale : 1.f;
    const float lse_shift = __log2f(LN2 * dsc);

    bf16x8 kf[KS], vf[KS];
    {
        const int j = j0 + 16 * w + r16;
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const size_t off = ((size_t)min(j, K - 1) * B + b) * a.ld_qkv + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = (j < K) ? ld_bf16x8(a.k + off + 32 * ks + 8 * g) : z;
            vf[ks] = (j < K) ? ld_bf16x8(a.v + off + 32 * ks + 8 * g) : z;
        }
    }
    int srcaddr[4];
    bool lower[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        srcaddr[reg] = ((lane & 48) | ((4 * g + reg - 1 - r16) & 15)) << 2;
        lower[reg] = r16 < 4 * g + reg;
    }
    // fragment addressing (elements): lane part r16 * DH + swizzled 16-byte chunk (block rows are multiples of 16, so
    // row & 7 == r16 & 7); the block's row offset is wave-uniform and mostly an immediate
    int foff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = r16 * DH + (((4 * ks + g) ^ (swz<DH>(r16) & (DH / 8 - 1))) << 3);
    f32x4 dk[DB], dv[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) dk[d] = dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // query tiles that see this kv tile: i >= j - M ; same_length: i < j + sshift
    int it_lo = max(0, j0 - M) >> 6;
    int it_hi = (T - 1) >> 6;
    if (a.same_length) it_hi = min(it_hi, (j0 + KCOLS - 1 + a.sshift - 1) >> 6);
    if (rst && j0 + KCOLS - 1 < M) it_hi = -1;          // whole tile is reset memory: no gradient
    if (it_hi < it_lo) it_hi = it_lo - 1;

    const unsigned qsb = (unsigned)B * HD * 2u, osb = (unsigned)B * a.ld_o * 2u, rdb = (unsigned)a.ld_rd * 2u;
    const srd_t srdQu = make_srd(a.qu2 + (size_t)b * HD + h * DH, ((size_t)(T - 1) * B * HD + DH) * 2);
    const srd_t srdQv = make_srd(a.qv2 + (size_t)b * HD + h * DH, ((size_t)(T - 1) * B * HD + DH) * 2);
    const srd_t srdO = make_srd(a.dout + (size_t)b * a.ld_o + h * DH, ((size_t)(T - 1) * B * a.ld_o + DH) * 2);
    const srd_t srdR = make_srd(a.rd + h * DH, ((size_t)(K - 1) * a.ld_rd + DH) * 2);
    Stager<64, DH, NTHR> stQu, stQv, stO, stR;
    stQu.init(qsb, tid);
    stQv.init(qsb, tid);
    stO.init(osb, tid);
    stR.init(rdb, tid);
    float plse = 0.f, pdl = 0.f;
    auto issue = [&](int it, int half) {          // band chunk `half` of tile `it`
        const int i0 = it * 64, dlo = i0 + M - j0 - (KCOLS - 1);
        stQu.load(srdQu, (unsigned)i0 * qsb);
        stQv.load(srdQv, (unsigned)i0 * qsb);
        stO.load(srdO, (unsigned)i0 * osb);
        stR.load(srdR, (unsigned)(dlo + 64 * half) * rdb);
        if (tid < 64) {
            const int i = min(i0 + tid, T - 1);
            plse = a.lse_in[((size_t)b * a.H + h) * T + i] * LOG2E - lse_shift;
            pdl = a.delta[((size_t)b * a.H + h) * T + i] / dsc;
        }
    };
    auto commit = [&](bf16* rdst) {
        stQu.store(sQu);
        stQv.store(sQv);
        stO.store(sdO);
        stR.store(rdst);
        if (tid < 64) { sLse[tid] = plse; sDl[tid] = pdl; }
    };
    if (it_lo <= it_hi) {
        // prologue: lower chunks of the first band (slot k at t = 0), then the tile with its top chunk
        const int dlo = it_lo * 64 + M - j0 - (KCOLS - 1);
#pragma unroll
        for (int kc = 0; kc < NCH - 1; ++kc) {
            stR.load(srdR, (unsigned)(dlo + 64 * kc) * rdb);
            stR.store(sR + kc * 64 * DH);
        }
        issue(it_lo, NCH - 1);
        commit(sR + (NCH - 1) * 64 * DH);
    }
    __syncthreads();
    for (int it = it_lo, t = 0; it <= it_hi; ++it, ++t) {
        const int i0 = it * 64;
        if (it < it_hi) issue(it + 1, NCH - 1);

        bf16x4 pb[4], dsb[4];     // per row block: P and dS'' for rows 16rb + 4g + reg, col r16
        const int jw_lo = j0 + 16 * w, jw_hi = jw_lo + 15;
        const bool need_mask = (jw_hi > i0 + M) || (a.same_length && jw_lo <= i0 + 63 - a.sshift) ||
                               (rst && jw_lo < M) || (i0 + 63 >= T) || (jw_hi >= K);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            f32x4 qr0 = {0.f, 0.f, 0.f, 0.f}, qr1 = {0.f, 0.f, 0.f, 0.f};
            const int base = 16 * (rb - w + NW - 1);      // band rows base .. base+31
            const bf16* r0 = sR + ring_row<NCH, 1>(base, t) * DH;
            const bf16* r1 = sR + ring_row<NCH, 1>(base + 16, t) * DH;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 qvf = ld_bf16x8(sQv + 16 * rb * DH + foff[ks]);
                s = mfma16(ld_bf16x8(sQu + 16 * rb * DH + foff[ks]), kf[ks], s);
                dp = mfma16(ld_bf16x8(sdO + 16 * rb * DH + foff[ks]), vf[ks], dp);
                qr0 = mfma16(qvf, ld_bf16x8(r0 + foff[ks]), qr0);
                qr1 = mfma16(qvf, ld_bf16x8(r1 + foff[ks]), qr1);
            }
            unsigned hw[4] = {0u, 0u, 0u, 0u};
            if (DROP) dl_.words((i0 >> 4) + rb, jw_lo >> 4, hw);
            const f32x4 lse4 = *(const f32x4*)&sLse[16 * rb + 4 * g], dl4 = *(const f32x4*)&sDl[16 * rb + 4 * g];
            // skewed band term: block select at the SOURCE lane (dest lane s < row  <=>  source lane t < row), one permute
            float sc[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) sc[reg] = s[reg] + bperm(srcaddr[reg], lower[reg] ? qr1[reg] : qr0[reg]);
            if (need_mask) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int ii = 16 * rb + 4 * g + reg;
                    if (is_masked(i0 + ii, jw_lo + r16, M, a.same_length, a.sshift, rst) || i0 + ii >= T) sc[reg] = -INFINITY;
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float p = __builtin_amdgcn_exp2f(sc[reg] - lse4[reg]);          // P' = P ln2 / (1-p)
                float pd = p, dpe = dp[reg];
                if (DROP) {
                    const bool keep = drop_keep16(hw, reg, a.drop_thr, thr_hi);
                    pd = keep ? p : 0.f;
                    dpe = keep ? dpe : 0.f;
                }
                pb[rb][reg] = f2bf(pd);
                dsb[rb][reg] = f2bf(p * (dpe - dl4[reg]));
            }
        }
        // dv += P^T dO ; dk += dS''^T qu2: k-slots e<4 -> ii = 32pp+4g+e, e>=4 -> ii = 32pp+16+4g+e-4
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            bf16x8 pa, da;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pa[e] = pb[2 * pp][e]; pa[4 + e] = pb[2 * pp + 1][e];
                da[e] = dsb[2 * pp][e]; da[4 + e] = dsb[2 * pp + 1][e];
            }
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                dv[d] = mfma16(pa, frag_tr_rm<DH>(sdO, 32 * pp + 4 * g, 32 * pp + 16 + 4 * g, 16 * d, r16), dv[d]);
                dk[d] = mfma16(da, frag_tr_rm<DH>(sQu, 32 * pp + 4 * g, 32 * pp + 16 + 4 * g, 16 * d, r16), dk[d]);
            }
        }
        __syncthreads();
        if (it < it_hi) commit(sR + ((t % NCH) << 6) * DH);      // the new top chunk replaces this tile's lowest chunk
        __syncthreads();
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int j = j0 + 16 * w + 4 * g + reg;
        if (j < K) {
            const size_t off = ((size_t)j * B + b) * a.ld_dqkv + h * DH;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                a.dk[off + 16 * d + r16] = f2bf(dk[d][reg]);
                a.dv[off + 16 * d + r16] = f2bf(dv[d][reg] * (1.f / LN2));
            }
        }
    }
    __syncthreads();
    }
}

// =============================================================================================
// Key-stationary backward from STORED probabilities (d_head 64).  The query-stationary kernel above recomputes
// P scale/(1-p) anyway; when the caller provides a scratch buffer it also writes it out (bf16, 2 KB per block of 16 rows
// x 64 keys, the lane order of the P^T image), and this kernel -- launched after it -- reads it back instead of
// recomputing (q+u).k, the band product with its skew, the masks and the exponentials: what remains is dP = dO.V^T,
// dS'' and the two products that consume P and dS''.  (Keeping the probabilities from the FORWARD pass was built and
// measured too: the 128-register forward kernel pays more for the extra stores than both backward kernels gain.)
// key-stationary: dk, dv (as relattn_bwd_kv_kernel: wave w owns key columns 16w..16w+15 of the 64-column tile).
template <bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void relattn_bwd_kv2_kernel(const AttnArgs a) {
    constexpr int DH = 64, NW = 4, KS = 2, DB = 4, KCOLS = 64, NTHR = 256;
    // (q+u) and dO tiles arrive by LDS-DMA into double buffers, the per-row scale / delta into double vectors: ONE barrier
    // per 64-query step, no staging registers, no tile writes from registers (the register-staged form needed two barriers
    // and 16 registers per lane in a kernel that lives at the 168-register / three-waves-per-SIMD edge)
    __shared__ __attribute__((aligned(1024))) bf16 sQu2[2][64 * DH];
    __shared__ __attribute__((aligned(1024))) bf16 sdO2[2][64 * DH];
    __shared__ __attribute__((aligned(16))) float sRs2[2][64], sDl2[2][64];

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
    const int T = a.T, M = a.M, B = a.B, K = T + M, HD = a.H * DH;
    const int NT = (K + KCOLS - 1) / KCOLS, NH = (NT + 1) / 2;      // key tiles j and NT-1-j back to back
    const int IB = (T + 15) >> 4, JT = (K + 63) >> 6;
    int jslot, h, b;
    tile_coords(NH, a.H, B, jslot, h, b);
    const bool rst = a.reset != nullptr && a.reset[b] != 0;
    // (dropout: the keep decisions arrive in the sign of the stored probabilities -- no hash in this kernel)
    // P' = P ln2/(1-p):  dS'' = P' (keep dP - delta (1-p)),  dV = [sum keep P' dO] / ln2   (see relattn_bwd_kv_kernel);
    // stored is P scale/(1-p)
    const float dsc = DROP ? a.drop_scale : 1.f;
    const float pmul = LN2 / a.scale;
    int foff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = r16 * DH + (((4 * ks + g) ^ (r16 & 7)) << 3);
    const unsigned qsb = (unsigned)B * HD * 2u, osb = (unsigned)B * a.ld_o * 2u;
    const srd_t srdQu = make_srd(a.qu2 + (size_t)b * HD + h * DH, ((size_t)(T - 1) * B * HD + DH) * 2);
    const srd_t srdO = make_srd(a.dout + (size_t)b * a.ld_o + h * DH, ((size_t)(T - 1) * B * a.ld_o + DH) * 2);
    // DMA piece j of wave w fills tile rows 16 w + 8 j + (lane >> 3); LDS slot (lane & 7) of a row holds logical 16-byte
    // chunk (lane & 7) ^ (row & 7) (the image frag / frag_tr_rm read: swz<64>)
    unsigned voffQ[2], voffO[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int R = 16 * w + 8 * j + (lane >> 3), c = (lane & 7) ^ (R & 7);
        voffQ[j] = (unsigned)R * qsb + (unsigned)c * 16u;
        voffO[j] = (unsigned)R * osb + (unsigned)c * 16u;
    }
    const unsigned ldsQ = (unsigned)(size_t)(LDS_AS bf16*)&sQu2[0][0] + (unsigned)(w * 2048);
    const unsigned ldsO = (unsigned)(size_t)(LDS_AS bf16*)&sdO2[0][0] + (unsigned)(w * 2048);
    const size_t bh = (size_t)b * a.H + h;

    for (int rep = 0; rep < 2; ++rep) {
    const int jt = rep == 0 ? jslot : NT - 1 - jslot;
    if (rep == 1 && jt <= jslot) break;
    const int j0 = jt * KCOLS, jw_lo = j0 + 16 * w;
    bf16x8 vf[KS];
    {
        const int j = jw_lo + r16;
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const size_t off = ((size_t)min(j, K - 1) * B + b) * a.ld_qkv + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vf[ks] = (j < K) ? ld_bf16x8(a.v + off + 32 * ks + 8 * g) : z;
    }
    f32x4 dk[DB], dv[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) dk[d] = dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // query tiles that see this kv tile: i >= j - M ; same_length: i < j + sshift
    int it_lo = max(0, j0 - M) >> 6;
    int it_hi = (T - 1) >> 6;
    if (a.same_length) it_hi = min(it_hi, (j0 + KCOLS - 1 + a.sshift - 1) >> 6);
    if (rst && j0 + KCOLS - 1 < M) it_hi = -1;          // whole tile is reset memory: no gradient
    if (it_hi < it_lo) it_hi = it_lo - 1;

    const bf16* pcol = a.pbuf + bh * IB * JT * 1024 + (size_t)jt * 1024 + pt_off(16 * w + r16, g);
    float prs = 0.f, pdl = 0.f;
    bf16x4 pun[4];
    auto issue = [&](int it, int buf) {
        const int i0 = it * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            lds_dma16s(srdQu, voffQ[j], (unsigned)i0 * qsb, ldsQ + (unsigned)(buf * 64 * DH * 2 + j * 1024));
            lds_dma16s(srdO, voffO[j], (unsigned)i0 * osb, ldsO + (unsigned)(buf * 64 * DH * 2 + j * 1024));
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)          // rows beyond the last 16-row block: clamp (their row scale is 0)
            pun[rb] = *(const bf16x4*)(pcol + (size_t)min((i0 >> 4) + rb, IB - 1) * JT * 1024);
        if (tid < 64) {
            const int i = i0 + tid;
            prs = i < T ? pmul : 0.f;          // rows past the end: their (clamped) block is somebody else's
            pdl = a.delta[bh * T + min(i, T - 1)] / dsc;
        }
    };
    auto commit = [&](int buf) {          // the per-row vectors of the step just requested (wave 0's lanes)
        if (tid < 64) { sRs2[buf][tid] = prs; sDl2[buf][tid] = pdl; }
    };
    if (it_lo <= it_hi) {
        issue(it_lo, 0);
        commit(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (the step body is instantiated for each of the two buffer sets: with the buffer index a compile-time constant every LDS
    //  address of the step is one per-lane base + an immediate -- a third of the step's vector instructions were address
    //  arithmetic on the runtime buffer index)
    auto step = [&](int it, auto curc) {
        constexpr int cur = decltype(curc)::value;
        const int i0 = it * 64;
        const bf16* sQu = sQu2[cur];
        const bf16* sdO = sdO2[cur];
        const float* sRs = sRs2[cur];
        const float* sDl = sDl2[cur];
        bf16x4 pu[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) pu[rb] = pun[rb];
        // (the other buffers were last read in the previous step, which every wave has left: the barrier below)
        if (it < it_hi) issue(it + 1, cur ^ 1);

        bf16x4 pb[4], dsb[4];     // per row block: P and dS'' for rows 16rb + 4g + reg, col r16
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dp = mfma16(ld_bf16x8(sdO + 16 * rb * DH + foff[ks]), vf[ks], dp);
            const f32x4 rs4 = *(const f32x4*)&sRs[16 * rb + 4 * g], dl4 = *(const f32x4*)&sDl[16 * rb + 4 * g];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float x = bf2f(pu[rb][reg]) * rs4[reg];          // +-P' = P ln2 / (1-p), negative: dropped (bwd_q)
                const float p = DROP ? __builtin_fabsf(x) : x;
                float pd = p, dpe = dp[reg];
                if (DROP) {
                    pd = __builtin_fmaxf(x, 0.f);
                    dpe = x > 0.f ? dpe : 0.f;          // (p == 0: either branch gives dS'' = 0)
                }
                pb[rb][reg] = f2bf(pd);
                dsb[rb][reg] = f2bf(p * (dpe - dl4[reg]));
            }
        }
        // dv += P^T dO ; dk += dS''^T qu2: k-slots e<4 -> ii = 32pp+4g+e, e>=4 -> ii = 32pp+16+4g+e-4
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            bf16x8 pa, da;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pa[e] = pb[2 * pp][e]; pa[4 + e] = pb[2 * pp + 1][e];
                da[e] = dsb[2 * pp][e]; da[4 + e] = dsb[2 * pp + 1][e];
            }
            // the four operand fragments of a feature-block pair (8 transpose reads) are requested before their MFMAs: left
            // alone hipcc emits read, read, wait, MFMA sixteen times and every MFMA pays a full LDS latency
#pragma unroll
            for (int dp2 = 0; dp2 < DB; dp2 += 2) {
                bf16x8 fo[2], fq[2];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    fo[d] = frag_tr_rm<DH>(sdO, 32 * pp + 4 * g, 32 * pp + 16 + 4 * g, 16 * (dp2 + d), r16);
                    fq[d] = frag_tr_rm<DH>(sQu, 32 * pp + 4 * g, 32 * pp + 16 + 4 * g, 16 * (dp2 + d), r16);
                }
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    dv[dp2 + d] = mfma16(pa, fo[d], dv[dp2 + d]);
                    dk[dp2 + d] = mfma16(da, fq[d], dk[dp2 + d]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        if (it < it_hi) commit(cur ^ 1);
        // the next step's tiles have landed (and its P values: they have had this step's arithmetic to arrive)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int it = it_lo; it <= it_hi; it += 2) {
        step(it, std::integral_constant<int, 0>{});
        if (it + 1 <= it_hi) step(it + 1, std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int j = jw_lo + 4 * g + reg;
        if (j < K) {
            const size_t off = ((size_t)j * B + b) * a.ld_dqkv + h * DH;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                a.dk[off + 16 * d + r16] = f2bf(dk[d][reg]);
                a.dv[off + 16 * d + r16] = f2bf(dv[d][reg] * (1.f / LN2));
            }
        }
    }
    __syncthreads();
    }
}

// delta[b,h,i] = sum_f dO[i,b,h,f] * O[i,b,h,f]
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ o, const bf16* __restrict__ dout,
                                                         int ld, float* __restrict__ delta, int T, int B,
                                                         int H, int DH) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);      // row m = i*B + b
    if (m >= T * B) return;
    const int lane = threadIdx.x & 63;
    const int i = m / B, b = m - i * B;
    const int lanes_per_head = DH / 8;
    for (int c0 = 0; c0 < H * DH; c0 += 512) {
        const int col = c0 + lane * 8;
        float s = 0.f;
        if (col < H * DH) {
            bf16x8 x = ld_bf16x8(o + (size_t)m * ld + col), y = ld_bf16x8(dout + (size_t)m * ld + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += bf2f(x[e]) * bf2f(y[e]);
        }
        for (int off = lanes_per_head >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (col < H * DH && (lane % lanes_per_head) == 0) {
            const int h = col / DH;
            delta[((size_t)b * H + h) * T + i] = s;
        }
    }
}

// dst[((b*H + h)*DH + f)*W + off + j] = src[(j*B + b)*ld + h*DH + f] (+ bias[h*DH+f]),  j in [0,J);
// every other column of the W-wide rows is zero.  64x64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_heads_kernel(const bf16* __restrict__ src, int ld,
                                                              const float* __restrict__ bias,
                                                              bf16* __restrict__ dst, int J, int B, int H,
                                                              int DH, int W, int off) {
    __shared__ bf16 t[64][66];
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int c0 = blockIdx.x * 64;            // destination column tile
    for (int f0 = 0; f0 < DH; f0 += 64) {
        for (int i = threadIdx.x; i < 4096; i += 256) {
            const int jj = i >> 6, f = i & 63;
            const int j = c0 + jj - off;
            float v = 0.f;
            if (j >= 0 && j < J && f0 + f < DH) {
                v = bf2f(src[((size_t)j * B + b) * ld + h * DH + f0 + f]);
                if (bias) v += bias[h * DH + f0 + f];
            }
            t[jj][f] = f2bf(v);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 4096; i += 256) {
            const int f = i >> 6, jj = i & 63;
            if (f0 + f < DH && c0 + jj < W)
                dst[(((size_t)b * H + h) * DH + f0 + f) * W + c0 + jj] = t[jj][f];
        }
        __syncthreads();
    }
}

}  // namespace

COMMU_DEFINE_SEED_SALT_SETTER(commu_seed_salt_relattn)

static void fill_common(AttnArgs& a, const commu_attn_desc* d) {
    a.u = d->r_w_bias; a.vb = d->r_r_bias; a.reset = d->reset;
    a.ld_qkv = d->ld_qkv; a.ld_rd = d->ld_rd; a.ld_o = d->ld_o;
    a.T = d->T; a.M = d->M; a.B = d->B; a.H = d->H;
    a.same_length = d->same_length; a.sshift = d->sshift; a.scale = d->scale;
    a.drop_seed = d->drop_seed;
    a.drop_thr = d->drop_p > 0.f ? (unsigned)((double)d->drop_p * 65536.0 + 0.5) : 0u;
    if (d->drop_p > 0.f && a.drop_thr == 0u) a.drop_thr = 1u;
    a.drop_scale = 1.f / (1.f - (float)a.drop_thr / 65536.f);      // scale by the exact keep probability
    a.q = (const bf16*)d->q; a.k = (const bf16*)d->k; a.v = (const bf16*)d->v; a.rd = (const bf16*)d->rd;
}

static bool fits_srd(const commu_attn_desc* d) {
    const size_t K = (size_t)d->T + d->M;
    return K * d->B * d->ld_qkv * 2 < 0xFFFF0000ull && K * d->ld_rd * 2 < 0xFFFF0000ull;
}

/* elements of the P scratch the backward pass may be given (commu_attn_bwd_desc.p_scratch) */
extern "C" long long commu_attn_p_scratch_elems(int T, int M, int B, int H) {
    return (long long)B * H * (2 * ((T + 31) / 32)) * ((T + M + 63) / 64) * 1024;          // (covers both block orders)
}

constexpr bool KV3_DEFAULT = true;           // the automatic choice (generation 0): relattn_kv3.hip since round 6 (-0.10 ms per step, 3 of 3
                                             // interleaved runs; equal in round 5, before the side-stream launches were re-balanced)
constexpr bool Q3_DEFAULT = false;           // generation 0 takes the 32x32-layout pair (relattn_q3.hip + relattn_kv3.hip, p_layout 2)
static int g_kv_gen = 0;
extern "C" int commu_attn_bwd_kv_generation(int gen) {
    const int prev = g_kv_gen;
    if (gen == 0 || gen == 2 || gen == 3 || gen == 4) g_kv_gen = gen;
    return prev;
}

static int g_fwd_gen = 0;
extern "C" int commu_attn_fwd_generation(int gen) {
    const int prev = g_fwd_gen;
    if (gen == 0 || gen == 2 || gen == 3) g_fwd_gen = gen;
    return prev;
}

extern "C" long long commu_attn_pf_bytes(int T, int M, int B, int H) {
    return (long long)B * H * ((T + 31) / 32) * ((T + M + 31) / 32) * PF_TILE_BYTES;
}

static int relattn_fwd_impl(const commu_attn_desc* d, void* out, float* lse, void* qu2, void* qv2, void* pf, hipStream_t stream);

extern "C" int commu_relattn_fwd(const commu_attn_desc* d, void* out, float* lse, void* qu2, void* qv2,
                                 hipStream_t stream) {
    return relattn_fwd_impl(d, out, lse, qu2, qv2, nullptr, stream);
}

extern "C" int commu_relattn_fwd_save(const commu_attn_desc* d, void* out, float* lse, void* qu2, void* qv2, void* pf,
                                      hipStream_t stream) {
    // (the tiles of one 32-query block are addressed through one buffer descriptor: 32-bit offsets)
    if (pf == nullptr || d->DH != 64 || (g_fwd_gen != 0 && g_fwd_gen != 3) ||
        (long long)((d->T + d->M + 31) / 32) * PF_TILE_BYTES >= 0x7FFF0000ll)
        return -22;
    return relattn_fwd_impl(d, out, lse, qu2, qv2, pf, stream);
}

static int relattn_fwd_impl(const commu_attn_desc* d, void* out, float* lse, void* qu2, void* qv2, void* pf,
                            hipStream_t stream) {
    if (d->T <= 0 || d->B <= 0) return 0;
    if ((d->ld_qkv % 8) || (d->ld_rd % 8) || !fits_srd(d) || ((qu2 == nullptr) != (qv2 == nullptr))) return -22;
    AttnArgs a = {};
    fill_common(a, d);
    a.out = (bf16*)out; a.lse = lse; a.qu2 = (bf16*)qu2; a.qv2 = (bf16*)qv2;
    a.pf = pf;
    // (the kernels are parametrised by waves per workgroup; 8-wave / 128-row tiles measured slower than 4-wave
    // tiles at every shape of this model, so only NW = 4 is instantiated)
    dim3 grid((((d->T + 63) / 64 + 1) / 2) * d->H * d->B);
    const bool drop = a.drop_thr != 0u;
    // d_head 64: third generation (relattn3.hip), with or without attention dropout -- every kernel regenerates the same mask
    // (commu_attn_fwd_generation(2) forces the 16x16 kernel)
    const int fwd_gen = g_fwd_gen;
    if (d->DH == 64 && (fwd_gen == 3 || fwd_gen == 0)) return launch_relattn_fwd3(a, stream);
    if (d->DH == 64 && fwd_gen == 2) {          // second-generation kernel: 128 query rows per workgroup
        dim3 grid2((((d->T + 127) / 128 + 1) / 2) * d->H * d->B);
        static const int stag = getenv("COMMU_ATTN_FWD_STAG") ? atoi(getenv("COMMU_ATTN_FWD_STAG")) : 0;
        if (stag == 1) {
            if (drop) COMMU_LAUNCH((relattn_fwd2_kernel<true, true, 2>), grid2, dim3(512), 0, stream, a);
            else COMMU_LAUNCH((relattn_fwd2_kernel<false, true, 2>), grid2, dim3(512), 0, stream, a);
        } else if (stag == 2) {          // lockstep at the staggered variant's occupancy (two waves per SIMD)
            if (drop) COMMU_LAUNCH((relattn_fwd2_kernel<true, false, 2>), grid2, dim3(512), 0, stream, a);
            else COMMU_LAUNCH((relattn_fwd2_kernel<false, false, 2>), grid2, dim3(512), 0, stream, a);
        } else {
            if (drop) COMMU_LAUNCH((relattn_fwd2_kernel<true, false>), grid2, dim3(512), 0, stream, a);
            else COMMU_LAUNCH((relattn_fwd2_kernel<false, false>), grid2, dim3(512), 0, stream, a);
        }
        COMMU_LAUNCH_CHECK();
        return 0;
    }
#define ATTN_FWD(DHV)                                                                              \
    {                                                                                              \
        if (drop) COMMU_LAUNCH((relattn_fwd_kernel<DHV, 4, true>), grid, dim3(256), 0, stream, a); \
        else COMMU_LAUNCH((relattn_fwd_kernel<DHV, 4, false>), grid, dim3(256), 0, stream, a);     \
    }
    if (d->DH == 64) ATTN_FWD(64)
    else if (d->DH == 32) ATTN_FWD(32)
    else return -22;
#undef ATTN_FWD
    COMMU_LAUNCH_CHECK();
    return 0;
}

// which: 1 = query-stationary kernel (dq_ac, dsk, du_part), 2 = key-stationary kernel (dk, dv), 3 = both
static int launch_relattn_bwd(const commu_attn_desc* d, const commu_attn_bwd_desc* e, int which, hipStream_t stream) {
    if (d->T <= 0 || d->B <= 0) return 0;
    const int K = d->T + d->M;
    if ((d->ld_qkv % 8) || (d->ld_rd % 8) || (e->ld_dqkv % 8) || !fits_srd(d) || e->ld_dsk < K) return -22;
    AttnArgs a = {};
    fill_common(a, d);
    a.dout = (const bf16*)e->dout; a.lse_in = e->lse; a.delta = e->delta;
    a.qu2 = (bf16*)e->qu2; a.qv2 = (bf16*)e->qv2;
    a.dq = (bf16*)e->dq_ac; a.dk = (bf16*)e->dk; a.dv = (bf16*)e->dv;
    a.dsk = (bf16*)e->dsk; a.du_part = e->du_part;
    a.ld_dqkv = e->ld_dqkv; a.ld_dsk = e->ld_dsk;
    a.dsk_wedge = e->dsk_wedge;
    a.dsk_tiled = e->dsk_tiled;
    a.pbuf = (bf16*)e->p_scratch;
    a.o_in = (const bf16*)e->o;
    a.pf = const_cast<void*>(e->pf);
    if (e->pf != nullptr && (e->p_scratch == nullptr || d->DH != 64)) return -22;
    if (e->p_scratch != nullptr && d->DH != 64) return -22;
    if (a.dsk_wedge > 0 && d->same_length) return -22;
    if ((e->ld_dsk % 8) || (a.dsk_tiled && ((e->ld_dsk % 128) || (((long long)d->T * d->B) % 64)))) return -22;
    if (e->du_rows != (d->T + 63) / 64) return -22;
    if ((size_t)d->T * d->B * e->ld_dsk * 2 >= 0x7FFF0000ull) return -22;          // the flush addresses one head's block with 32-bit offsets
    dim3 gq((((d->T + 63) / 64 + 1) / 2) * d->H * d->B), gk((((K + 63) / 64 + 1) / 2) * d->H * d->B);
    const bool drop = a.drop_thr != 0u;
    if (a.pbuf != nullptr) {          // the query-stationary kernel stores P, the key-stationary one reads it back
        // (commu_attn_bwd_kv_generation: 3 = relattn_kv3.hip, 32 keys per wave on the 32x32 MFMA; the two launches of a
        //  backward pass must see the same setting -- it fixes the block order of the P scratch)
        // generation 4: the query-stationary kernel of relattn_q3.hip (32 query rows per wave on the 32x32 MFMA, transposed
        // scores) with relattn_kv3.hip reading its block order (p_layout 2)
        if ((g_kv_gen == 4 || (Q3_DEFAULT && g_kv_gen == 0)) && relattn_bwd_q3_takes(a)) {
            a.p_layout = 2;
            if (which & 1) {
                const int rc = launch_relattn_bwd_q3(a, stream);
                if (rc != 0) return rc;
            }
            if (which & 2) launch_relattn_bwd_kv3(a, stream);
            COMMU_LAUNCH_CHECK();
            return 0;
        }
        // generation 0 (the default): the 16x16 query-stationary kernel with its whole-line P stores (p_layout 0) and
        // relattn_kv3.hip reading that order; 3: the same pair with the 32x32 block order (p_layout 1); 2: the 16x16 pair
        const bool kv3 = KV3_DEFAULT ? g_kv_gen != 2 : g_kv_gen == 3;
        a.p_layout = (kv3 && g_kv_gen == 3) ? 1 : 0;
        const bool fromp = a.pf != nullptr;          // probabilities saved by the forward pass: no score recomputation
        if (drop) {
            if ((which & 1) && fromp) COMMU_LAUNCH((relattn_bwd_q_kernel<64, 4, true, true>), gq, dim3(256), 0, stream, a);
            if ((which & 1) && !fromp) COMMU_LAUNCH((relattn_bwd_q_kernel<64, 4, true>), gq, dim3(256), 0, stream, a);
            if ((which & 2) && !kv3) COMMU_LAUNCH((relattn_bwd_kv2_kernel<true>), gk, dim3(256), 0, stream, a);
        } else {
            if ((which & 1) && fromp) COMMU_LAUNCH((relattn_bwd_q_kernel<64, 4, false, true>), gq, dim3(256), 0, stream, a);
            if ((which & 1) && !fromp) COMMU_LAUNCH((relattn_bwd_q_kernel<64, 4, false>), gq, dim3(256), 0, stream, a);
            if ((which & 2) && !kv3) COMMU_LAUNCH((relattn_bwd_kv2_kernel<false>), gk, dim3(256), 0, stream, a);
        }
        if ((which & 2) && kv3) launch_relattn_bwd_kv3(a, stream);
        COMMU_LAUNCH_CHECK();
        return 0;
    }
#define ATTN_BWD(DHV)                                                                                                  \
    {                                                                                                                  \
        if (drop) {                                                                                                    \
            if (which & 1) COMMU_LAUNCH((relattn_bwd_q_kernel<DHV, 4, true>), gq, dim3(256), 0, stream, a);            \
            if (which & 2) COMMU_LAUNCH((relattn_bwd_kv_kernel<DHV, 4, true>), gk, dim3(256), 0, stream, a);           \
        } else {                                                                                                       \
            if (which & 1) COMMU_LAUNCH((relattn_bwd_q_kernel<DHV, 4, false>), gq, dim3(256), 0, stream, a);           \
            if (which & 2) COMMU_LAUNCH((relattn_bwd_kv_kernel<DHV, 4, false>), gk, dim3(256), 0, stream, a);          \
        }                                                                                                              \
    }
    if (d->DH == 64) ATTN_BWD(64)
    else if (d->DH == 32) ATTN_BWD(32)
    else return -22;
#undef ATTN_BWD
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_relattn_bwd(const commu_attn_desc* d, const commu_attn_bwd_desc* e, hipStream_t stream) {
    return launch_relattn_bwd(d, e, 3, stream);
}
extern "C" int commu_relattn_bwd_q(const commu_attn_desc* d, const commu_attn_bwd_desc* e, hipStream_t stream) {
    return launch_relattn_bwd(d, e, 1, stream);
}
extern "C" int commu_relattn_bwd_kv(const commu_attn_desc* d, const commu_attn_bwd_desc* e, hipStream_t stream) {
    return launch_relattn_bwd(d, e, 2, stream);
}

extern "C" int commu_attn_delta(const void* o, const void* dout, int ld, float* delta, int T, int B, int H,
                                int DH, hipStream_t stream) {
    if (T * B <= 0) return 0;
    if ((DH != 32 && DH != 64 && DH != 128) || (ld % 8)) return -22;
    COMMU_LAUNCH(attn_delta_kernel, dim3((T * B + 3) / 4), dim3(256), 0, stream, (const bf16*)o,
                       (const bf16*)dout, ld, delta, T, B, H, DH);
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_transpose_heads(const void* src, int ld, const float* bias, void* dst, int J, int B, int H,
                                     int DH, int W, int off, hipStream_t stream) {
    if (B * H <= 0 || W <= 0) return 0;
    COMMU_LAUNCH(transpose_heads_kernel, dim3((W + 63) / 64, B * H), dim3(256), 0, stream,
                       (const bf16*)src, ld, bias, (bf16*)dst, J, B, H, DH, W, off);
    COMMU_LAUNCH_CHECK();
    return 0;
}

/* query-tile rows the backward kernels use for a given T (du_part has ceil(T / rows) tiles per batch entry) */
extern "C" int commu_attn_bwd_qrows(int T) { (void)T; return 64; }
