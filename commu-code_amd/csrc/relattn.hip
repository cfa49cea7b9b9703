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
// (b,h) matrix is `base + (j*B + b)*ld + h*DH`.  Operands whose contraction index is the ROW
// index (V in P.V, K in dS.K, Rd in dQR.Rd, Q/dO in the dK/dV products) are read from
// pre-transposed copies [b][h][f][j] made by commu_transpose_heads, so every LDS image is a
// plain row-major tile with an XOR swizzle and every fragment is one 16-byte (or 8-byte) read.
//
// Kernels: relattn_fwd (q-stationary), relattn_bwd_q (q-stationary: dq, du/dv partials, the
// skewed dS band for dR), relattn_bwd_kv (kv-stationary: dk, dv), attn_delta, transpose_heads.
#include "common.cuh"
#include "commu_hip.h"

namespace {

struct AttnArgs {
    const bf16* q;      // rows i in [0,T)   : q  + (i*B+b)*ld_qkv + h*DH
    const bf16* k;      // rows j in [0,K)   : k  + (j*B+b)*ld_qkv + h*DH
    const bf16* v;
    const bf16* kt;     // [B][H][DH][Jpad]  (bwd_q)
    const bf16* vt;     // [B][H][DH][Jpad]  (fwd)
    const bf16* rd;     // [K][H*DH] distance-indexed
    const bf16* rdt;    // [H][DH][Wr], entry [f][128 + sft + d] = Rd[d][f]   (bwd_q)
    const bf16* qut;    // [B][H][DH][Tpad]  (q+u)^T  (bwd_kv)
    const bf16* dot;    // [B][H][DH][Tpad]  dO^T     (bwd_kv)
    const float* u;     // r_w_bias [H][DH]
    const float* vb;    // r_r_bias [H][DH]
    const unsigned char* reset;   // [B] or null
    const bf16* o;      // forward output (bwd)
    const bf16* dout;   // dO rows like q, ld_o
    const float* lse_in;
    const float* delta;
    bf16* out;          // [T*B][ld_o]
    float* lse;         // [B][H][T]
    bf16* dq;           // rows like q, ld_dqkv
    bf16* dk;
    bf16* dv;
    bf16* qv_out;       // [T*B][H*DH]  (q+v), for the dR GEMM
    bf16* dsk;          // [H][T*B][ld_dsk]  skewed dS (index d), for the dR GEMM
    float* du_part;     // [B*QT][H*DH]
    float* dvb_part;
    int ld_qkv, ld_rd, ld_o, ld_dqkv, ld_dsk;
    int T, M, B, H, Jpad, Tpad, Wr, sft;
    int same_length, sshift;
    float scale;
};

template <int COLS>
__device__ __forceinline__ int swz(int row) {
    constexpr int CH = COLS / 8;
    if (CH >= 16) return row & 15;
    if (CH == 8) return row & 7;
    return ((row >> 3) & 1) * 3;      // 64-byte rows (see gemm.hip swz64)
}

// stage ROWS x COLS (bf16) into a swizzled row-major LDS tile; row r comes from
// src + r*rstride; rows outside [vlo, vhi) are zero-filled.
template <int ROWS, int COLS>
__device__ __forceinline__ void stage(bf16* dst, const bf16* src, size_t rstride, int vlo, int vhi, int tid) {
    constexpr int CH = COLS / 8;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = tid; i < ROWS * CH; i += 256) {
        const int r = i / CH, c = i % CH;
        bf16x8 v = z;
        if (r >= vlo && r < vhi) v = ld_bf16x8(src + (ptrdiff_t)r * (ptrdiff_t)rstride + c * 8);
        st_bf16x8(dst + r * COLS + ((c ^ (swz<COLS>(r) & (CH - 1))) << 3), v);
    }
}
// same, adding a per-column fp32 bias (q + u / q + v)
template <int ROWS, int COLS>
__device__ __forceinline__ void stage_bias(bf16* dst, const bf16* src, size_t rstride, int vlo, int vhi,
                                           const float* bias, int tid) {
    constexpr int CH = COLS / 8;
#pragma unroll
    for (int i = tid; i < ROWS * CH; i += 256) {
        const int r = i / CH, c = i % CH;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r >= vlo && r < vhi) {
            bf16x8 raw = ld_bf16x8(src + (ptrdiff_t)r * (ptrdiff_t)rstride + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(raw[e]) + bias[c * 8 + e]);
        }
        st_bf16x8(dst + r * COLS + ((c ^ (swz<COLS>(r) & (CH - 1))) << 3), v);
    }
}
template <int COLS>
__device__ __forceinline__ bf16x8 frag(const bf16* tile, int row, int chunk) {
    constexpr int CH = COLS / 8;
    return ld_bf16x8(tile + row * COLS + ((chunk ^ (swz<COLS>(row) & (CH - 1))) << 3));
}
// 4 consecutive elements starting at column col (col % 4 == 0)
template <int COLS>
__device__ __forceinline__ bf16x4 frag4(const bf16* tile, int row, int col) {
    constexpr int CH = COLS / 8;
    const int chunk = col >> 3;
    return *(const bf16x4*)(tile + row * COLS + ((chunk ^ (swz<COLS>(row) & (CH - 1))) << 3) + (col & 7));
}

__device__ __forceinline__ float bperm(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

__device__ __forceinline__ bool is_masked(int i, int j, int M, int same_length, int sshift, bool rst) {
    return (j > i + M) || (same_length && j <= i - sshift) || (rst && j < M);
}

// kv tile range visible from query rows [i0, i0+63]
__device__ __forceinline__ void kv_range(const AttnArgs& a, int i0, bool rst, int& jt_lo, int& jt_hi) {
    const int K = a.T + a.M;
    int jlo = rst ? a.M : 0;
    if (a.same_length) jlo = max(jlo, i0 - a.sshift + 1);
    jlo = max(jlo, 0);
    const int jhi = min(K - 1, i0 + 63 + a.M);
    jt_lo = jlo >> 6;
    jt_hi = jhi >> 6;
}

constexpr int PP = 72;    // pitch of the per-wave P tile [16][64] (+8 pad: 144-byte rows)
constexpr int PP2 = 104;  // pitch of the per-wave dQR tile [16][96] (+8)

// =============================================================================================
template <int DH>
__global__ __launch_bounds__(256) void relattn_fwd_kernel(const AttnArgs a) {
    constexpr int KS = DH / 32, DB = DH / 16;
    __shared__ __attribute__((aligned(16))) bf16 sK[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sVt[DH * 64];
    __shared__ __attribute__((aligned(16))) bf16 sR[128 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sP[4 * 16 * PP];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int qt = gridDim.x - 1 - blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int i0 = qt * 64, T = a.T, M = a.M, B = a.B, K = T + M;
    const bool rst = a.reset != nullptr && a.reset[b] != 0;
    const size_t rs = (size_t)B * a.ld_qkv;

    bf16x8 qu[KS], qv[KS];
    {
        const int iq = min(i0 + 16 * w + r16, T - 1);
        const bf16* qp = a.q + ((size_t)iq * B + b) * a.ld_qkv + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 raw = ld_bf16x8(qp + 32 * ks + 8 * g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int f = h * DH + 32 * ks + 8 * g + e;
                const float x = bf2f(raw[e]);
                qu[ks][e] = f2bf(x + a.u[f]);
                qv[ks][e] = f2bf(x + a.vb[f]);
            }
        }
    }
    int srcaddr[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) srcaddr[reg] = ((lane & 48) | ((4 * g + reg - 1 - r16) & 15)) << 2;

    f32x4 o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrow[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    float lpart[4] = {0.f, 0.f, 0.f, 0.f};

    int jt_lo, jt_hi;
    kv_range(a, i0, rst, jt_lo, jt_hi);
    bf16* myP = sP + w * 16 * PP;
    for (int jt = jt_lo; jt <= jt_hi; ++jt) {
        const int j0 = jt * 64;
        const int dlo = i0 + M - j0 - 63;
        __syncthreads();
        stage<64, DH>(sK, a.k + ((size_t)j0 * B + b) * a.ld_qkv + h * DH, rs, 0, K - j0, tid);
        stage<DH, 64>(sVt, a.vt + (((size_t)b * a.H + h) * DH) * a.Jpad + j0, (size_t)a.Jpad, 0, DH, tid);
        stage<128, DH>(sR, a.rd + (ptrdiff_t)dlo * a.ld_rd + h * DH, (size_t)a.ld_rd, -dlo, K - dlo, tid);
        __syncthreads();

        f32x4 s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[c] = mfma16(qu[ks], frag<DH>(sK, 16 * c + r16, 4 * ks + g), s[c]);
        }
        f32x4 qr[5];
#pragma unroll
        for (int blk = 0; blk < 5; ++blk) {
            qr[blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qr[blk] = mfma16(qv[ks], frag<DH>(sR, 16 * w + 16 * blk + r16, 4 * ks + g), qr[blk]);
        }
        // skew + scale + mask
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float pm[5];
#pragma unroll
            for (int blk = 0; blk < 5; ++blk) pm[blk] = bperm(srcaddr[reg], qr[blk][reg]);
            const int row = 4 * g + reg;
            const int i = i0 + 16 * w + row;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float bd = (r16 < row) ? pm[4 - c] : pm[3 - c];
                const int j = j0 + 16 * c + r16;
                const float sc = (s[c][reg] + bd) * a.scale;
                s[c][reg] = is_masked(i, j, M, a.same_length, a.sshift, rst) ? -INFINITY : sc;
            }
        }
        // online softmax
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float mx = fmaxf(fmaxf(s[0][reg], s[1][reg]), fmaxf(s[2][reg], s[3][reg]));
            mx = row16_max(mx);
            const float mnew = fmaxf(mrow[reg], mx);
            const float alpha = __expf(mrow[reg] - mnew);
            mrow[reg] = mnew;
            float ps = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float p = __expf(s[c][reg] - mnew);
                ps += p;
                myP[(4 * g + reg) * PP + 16 * c + r16] = f2bf(p);
            }
            lpart[reg] = lpart[reg] * alpha + ps;
#pragma unroll
            for (int d = 0; d < DB; ++d) o[d][reg] *= alpha;
        }
        __builtin_amdgcn_wave_barrier();
        bf16x8 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) pf[ks] = ld_bf16x8(myP + r16 * PP + 32 * ks + 8 * g);
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) o[d] = mfma16(pf[ks], frag<64>(sVt, 16 * d + r16, 4 * ks + g), o[d]);
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const float l = row16_sum(lpart[reg]);
        const float inv = 1.f / l;
        const int i = i0 + 16 * w + 4 * g + reg;
        if (i < T) {
            bf16* op = a.out + ((size_t)i * B + b) * a.ld_o + h * DH;
#pragma unroll
            for (int d = 0; d < DB; ++d) op[16 * d + r16] = f2bf(o[d][reg] * inv);
            if (r16 == 0) a.lse[((size_t)b * a.H + h) * T + i] = mrow[reg] + __logf(l);
        }
    }
}

// =============================================================================================
// backward, q-stationary: dq (= dq_ac + dq_bd), per-block column sums of dq_ac / dq_bd (du, dv
// bias grads), (q+v) copy and the skewed dS band dSk[i][d] for the dR GEMM.
template <int DH>
__global__ __launch_bounds__(256) void relattn_bwd_q_kernel(const AttnArgs a) {
    constexpr int KS = DH / 32, DB = DH / 16;
    __shared__ __attribute__((aligned(16))) bf16 sK[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sKt[DH * 64];
    __shared__ __attribute__((aligned(16))) bf16 sV[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sR[128 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sRt[DH * 128];
    __shared__ __attribute__((aligned(16))) bf16 sP[4 * 16 * PP];
    __shared__ __attribute__((aligned(16))) bf16 sP2[4 * 16 * PP2];
    __shared__ float red[2][4][DH];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int qt = gridDim.x - 1 - blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int i0 = qt * 64, T = a.T, M = a.M, B = a.B, K = T + M;
    const bool rst = a.reset != nullptr && a.reset[b] != 0;
    const size_t rs = (size_t)B * a.ld_qkv;

    bf16x8 qu[KS], qv[KS], dof[KS];
    {
        const int irow = i0 + 16 * w + r16;
        const int iq = min(irow, T - 1);
        const bf16* qp = a.q + ((size_t)iq * B + b) * a.ld_qkv + h * DH;
        const bf16* dop = a.dout + ((size_t)iq * B + b) * a.ld_o + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 raw = ld_bf16x8(qp + 32 * ks + 8 * g);
            dof[ks] = ld_bf16x8(dop + 32 * ks + 8 * g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int f = h * DH + 32 * ks + 8 * g + e;
                const float x = bf2f(raw[e]);
                qu[ks][e] = f2bf(x + a.u[f]);
                qv[ks][e] = f2bf(x + a.vb[f]);
            }
            if (irow < T)
                st_bf16x8(a.qv_out + ((size_t)irow * B + b) * (a.H * DH) + h * DH + 32 * ks + 8 * g, qv[ks]);
        }
    }
    float lse[4], dl[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int i = min(i0 + 16 * w + 4 * g + reg, T - 1);
        lse[reg] = a.lse_in[((size_t)b * a.H + h) * T + i];
        dl[reg] = a.delta[((size_t)b * a.H + h) * T + i];
    }
    int srcaddr[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) srcaddr[reg] = ((lane & 48) | ((4 * g + reg - 1 - r16) & 15)) << 2;

    f32x4 dq_ac[DB], dq_bd[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) dq_ac[d] = dq_bd[d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16* myP = sP + w * 16 * PP;
    bf16* myP2 = sP2 + w * 16 * PP2;
    // zero the k-padding columns 80..95 of the dQR tile once
    for (int i = lane; i < 16 * 16; i += 64) myP2[(i >> 4) * PP2 + 80 + (i & 15)] = f2bf(0.f);

    int jt_lo, jt_hi;
    kv_range(a, i0, rst, jt_lo, jt_hi);
    const size_t mrow0 = (size_t)a.T * B;        // rows per head in dSk
    for (int jt = jt_lo; jt <= jt_hi; ++jt) {
        const int j0 = jt * 64;
        const int dlo = i0 + M - j0 - 63;
        __syncthreads();
        stage<64, DH>(sK, a.k + ((size_t)j0 * B + b) * a.ld_qkv + h * DH, rs, 0, K - j0, tid);
        stage<64, DH>(sV, a.v + ((size_t)j0 * B + b) * a.ld_qkv + h * DH, rs, 0, K - j0, tid);
        stage<DH, 64>(sKt, a.kt + (((size_t)b * a.H + h) * DH) * a.Jpad + j0, (size_t)a.Jpad, 0, DH, tid);
        stage<128, DH>(sR, a.rd + (ptrdiff_t)dlo * a.ld_rd + h * DH, (size_t)a.ld_rd, -dlo, K - dlo, tid);
        stage<DH, 128>(sRt, a.rdt + ((size_t)h * DH) * a.Wr + (128 + a.sft + dlo), (size_t)a.Wr, 0, DH, tid);
        __syncthreads();

        f32x4 s[4], dp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s[c] = dp[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s[c] = mfma16(qu[ks], frag<DH>(sK, 16 * c + r16, 4 * ks + g), s[c]);
                dp[c] = mfma16(dof[ks], frag<DH>(sV, 16 * c + r16, 4 * ks + g), dp[c]);
            }
        }
        f32x4 qr[5];
#pragma unroll
        for (int blk = 0; blk < 5; ++blk) {
            qr[blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qr[blk] = mfma16(qv[ks], frag<DH>(sR, 16 * w + 16 * blk + r16, 4 * ks + g), qr[blk]);
        }
        // dS (in s[c][reg]) then its A-layout copy + the un-skewed band
        const int dlo_w = dlo + 16 * w;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float pm[5];
#pragma unroll
            for (int blk = 0; blk < 5; ++blk) pm[blk] = bperm(srcaddr[reg], qr[blk][reg]);
            const int row = 4 * g + reg;
            const int i = i0 + 16 * w + row;
            float ds[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float bd = (r16 < row) ? pm[4 - c] : pm[3 - c];
                const int j = j0 + 16 * c + r16;
                const float sc = (s[c][reg] + bd) * a.scale;
                const bool msk = is_masked(i, j, M, a.same_length, a.sshift, rst) || (i >= T);
                const float p = msk ? 0.f : __expf(sc - lse[reg]);
                ds[c] = p * (dp[c][reg] - dl[reg]) * a.scale;
                myP[row * PP + 16 * c + r16] = f2bf(ds[c]);
            }
            float dsp[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) dsp[c] = bperm(srcaddr[reg], ds[c]);
#pragma unroll
            for (int blk = 0; blk < 5; ++blk) {
                const float hi = (blk >= 1) ? dsp[4 - blk] : 0.f;      // c = 4-blk valid for blk>=1
                const float lo = (blk <= 3) ? dsp[3 - blk] : 0.f;      // c = 3-blk valid for blk<=3
                const float val = (r16 < row) ? hi : lo;
                const bf16 vb = f2bf(val);
                myP2[row * PP2 + 16 * blk + r16] = vb;
                const int bidx = 16 * blk + r16;
                const int jj = row + 63 - bidx;
                const int d = dlo_w + bidx;
                if (jj >= 0 && jj <= 63 && d >= 0 && d < K && i < T)
                    a.dsk[((size_t)h * mrow0 + (size_t)i * B + b) * a.ld_dsk + d] = vb;
            }
        }
        __builtin_amdgcn_wave_barrier();
        bf16x8 pf[2], pf2[3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) pf[ks] = ld_bf16x8(myP + r16 * PP + 32 * ks + 8 * g);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) pf2[ks] = ld_bf16x8(myP2 + r16 * PP2 + 32 * ks + 8 * g);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                dq_ac[d] = mfma16(pf[ks], frag<64>(sKt, 16 * d + r16, 4 * ks + g), dq_ac[d]);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
                dq_bd[d] = mfma16(pf2[ks], frag<128>(sRt, 16 * d + r16, min(2 * w + 4 * ks + g, 15)), dq_bd[d]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // outputs
#pragma unroll
    for (int d = 0; d < DB; ++d) {
        float ca = 0.f, cb = 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = i0 + 16 * w + 4 * g + reg;
            if (i < T) {
                a.dq[((size_t)i * B + b) * a.ld_dqkv + h * DH + 16 * d + r16] = f2bf(dq_ac[d][reg] + dq_bd[d][reg]);
                ca += dq_ac[d][reg];
                cb += dq_bd[d][reg];
            }
        }
        ca += __shfl_xor(ca, 16, 64); ca += __shfl_xor(ca, 32, 64);
        cb += __shfl_xor(cb, 16, 64); cb += __shfl_xor(cb, 32, 64);
        if (g == 0) { red[0][w][16 * d + r16] = ca; red[1][w][16 * d + r16] = cb; }
    }
    __syncthreads();
    if (tid < DH) {
        const size_t off = ((size_t)b * gridDim.x + qt) * (a.H * DH) + h * DH + tid;
        a.du_part[off] = red[0][0][tid] + red[0][1][tid] + red[0][2][tid] + red[0][3][tid];
        a.dvb_part[off] = red[1][0][tid] + red[1][1][tid] + red[1][2][tid] + red[1][3][tid];
    }
}

// =============================================================================================
// backward, kv-stationary: dk, dv.  Wave w owns kv columns 16w..16w+15 of the tile; S, dP are
// held as [64 q rows x 16 kv cols] in C layout, which IS the A-operand layout of S^T / dS^T.
template <int DH>
__global__ __launch_bounds__(256) void relattn_bwd_kv_kernel(const AttnArgs a) {
    constexpr int KS = DH / 32, DB = DH / 16;
    __shared__ __attribute__((aligned(16))) bf16 sQu[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sQv[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sQut[DH * 64];
    __shared__ __attribute__((aligned(16))) bf16 sdO[64 * DH];
    __shared__ __attribute__((aligned(16))) bf16 sdOt[DH * 64];
    __shared__ __attribute__((aligned(16))) bf16 sR[128 * DH];
    __shared__ float sLse[64], sDl[64];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int jt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int j0 = jt * 64, T = a.T, M = a.M, B = a.B, K = T + M;
    const bool rst = a.reset != nullptr && a.reset[b] != 0;
    const size_t rs = (size_t)B * a.ld_qkv;

    bf16x8 kf[KS], vf[KS];
    {
        const int j = j0 + 16 * w + r16;
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const bf16* kp = a.k + ((size_t)min(j, K - 1) * B + b) * a.ld_qkv + h * DH;
        const bf16* vp = a.v + ((size_t)min(j, K - 1) * B + b) * a.ld_qkv + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = (j < K) ? ld_bf16x8(kp + 32 * ks + 8 * g) : z;
            vf[ks] = (j < K) ? ld_bf16x8(vp + 32 * ks + 8 * g) : z;
        }
    }
    int srcaddr[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) srcaddr[reg] = ((lane & 48) | ((4 * g + reg - 1 - r16) & 15)) << 2;

    f32x4 dk[DB], dv[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) dk[d] = dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // query tiles that see this kv tile: i >= j - M ; same_length: i < j + sshift
    int it_lo = max(0, j0 - M) >> 6;
    int it_hi = (T - 1) >> 6;
    if (a.same_length) it_hi = min(it_hi, (j0 + 63 + a.sshift - 1) >> 6);
    if (rst && j0 + 63 < M) it_hi = -1;          // whole tile is reset memory: no gradient
    if (it_hi < it_lo) it_hi = it_lo - 1;

    for (int it = it_lo; it <= it_hi; ++it) {
        const int i0 = it * 64;
        const int dlo = i0 + M - j0 - 63;
        __syncthreads();
        const bf16* qbase = a.q + ((size_t)i0 * B + b) * a.ld_qkv + h * DH;
        stage_bias<64, DH>(sQu, qbase, rs, 0, T - i0, a.u + h * DH, tid);
        stage_bias<64, DH>(sQv, qbase, rs, 0, T - i0, a.vb + h * DH, tid);
        stage<64, DH>(sdO, a.dout + ((size_t)i0 * B + b) * a.ld_o + h * DH, (size_t)B * a.ld_o, 0, T - i0, tid);
        stage<DH, 64>(sQut, a.qut + (((size_t)b * a.H + h) * DH) * a.Tpad + i0, (size_t)a.Tpad, 0, DH, tid);
        stage<DH, 64>(sdOt, a.dot + (((size_t)b * a.H + h) * DH) * a.Tpad + i0, (size_t)a.Tpad, 0, DH, tid);
        stage<128, DH>(sR, a.rd + (ptrdiff_t)dlo * a.ld_rd + h * DH, (size_t)a.ld_rd, -dlo, K - dlo, tid);
        if (tid < 64) {
            const int i = min(i0 + tid, T - 1);
            sLse[tid] = a.lse_in[((size_t)b * a.H + h) * T + i];
            sDl[tid] = a.delta[((size_t)b * a.H + h) * T + i];
        }
        __syncthreads();

        bf16x4 pb[4], dsb[4];     // per row block: P and dS for rows 16rb + 4g + reg, col r16
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            f32x4 qr0 = {0.f, 0.f, 0.f, 0.f}, qr1 = {0.f, 0.f, 0.f, 0.f};
            const int base = 16 * (rb - w) + 48;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 quf = frag<DH>(sQu, 16 * rb + r16, 4 * ks + g);
                const bf16x8 qvf = frag<DH>(sQv, 16 * rb + r16, 4 * ks + g);
                s = mfma16(quf, kf[ks], s);
                dp = mfma16(frag<DH>(sdO, 16 * rb + r16, 4 * ks + g), vf[ks], dp);
                qr0 = mfma16(qvf, frag<DH>(sR, base + r16, 4 * ks + g), qr0);
                qr1 = mfma16(qvf, frag<DH>(sR, base + 16 + r16, 4 * ks + g), qr1);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 4 * g + reg;
                const float p0 = bperm(srcaddr[reg], qr0[reg]);
                const float p1 = bperm(srcaddr[reg], qr1[reg]);
                const float bd = (r16 < row) ? p1 : p0;
                const int ii = 16 * rb + row;
                const int i = i0 + ii, j = j0 + 16 * w + r16;
                const float sc = (s[reg] + bd) * a.scale;
                const bool msk = is_masked(i, j, M, a.same_length, a.sshift, rst) || (i >= T);
                const float p = msk ? 0.f : __expf(sc - sLse[ii]);
                pb[rb][reg] = f2bf(p);
                dsb[rb][reg] = f2bf(p * (dp[reg] - sDl[ii]) * a.scale);
            }
        }
        // dv += P^T dO ; dk += dS^T (q+u): k-slots e<4 -> ii = 32pp+4g+e, e>=4 -> ii = 32pp+16+4g+e-4
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
                const bf16x4 x0 = frag4<64>(sdOt, 16 * d + r16, 32 * pp + 4 * g);
                const bf16x4 x1 = frag4<64>(sdOt, 16 * d + r16, 32 * pp + 16 + 4 * g);
                const bf16x8 xb = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                dv[d] = mfma16(pa, xb, dv[d]);
                const bf16x4 y0 = frag4<64>(sQut, 16 * d + r16, 32 * pp + 4 * g);
                const bf16x4 y1 = frag4<64>(sQut, 16 * d + r16, 32 * pp + 16 + 4 * g);
                const bf16x8 yb = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
                dk[d] = mfma16(da, yb, dk[d]);
            }
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int j = j0 + 16 * w + 4 * g + reg;
        if (j < K) {
            const size_t off = ((size_t)j * B + b) * a.ld_dqkv + h * DH;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                a.dk[off + 16 * d + r16] = f2bf(dk[d][reg]);
                a.dv[off + 16 * d + r16] = f2bf(dv[d][reg]);
            }
        }
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

static void fill_common(AttnArgs& a, const commu_attn_desc* d) {
    a.u = d->r_w_bias; a.vb = d->r_r_bias; a.reset = d->reset;
    a.ld_qkv = d->ld_qkv; a.ld_rd = d->ld_rd; a.ld_o = d->ld_o;
    a.T = d->T; a.M = d->M; a.B = d->B; a.H = d->H;
    a.same_length = d->same_length; a.sshift = d->sshift; a.scale = d->scale;
    a.q = (const bf16*)d->q; a.k = (const bf16*)d->k; a.v = (const bf16*)d->v; a.rd = (const bf16*)d->rd;
}

extern "C" int commu_relattn_fwd(const commu_attn_desc* d, const void* vt, int Jpad, void* out, float* lse,
                                 hipStream_t stream) {
    if (d->T <= 0 || d->B <= 0) return 0;
    if ((d->ld_qkv % 8) || (d->ld_rd % 8) || (Jpad % 64) || Jpad < ((d->T + d->M + 63) / 64) * 64) return -22;
    AttnArgs a = {};
    fill_common(a, d);
    a.vt = (const bf16*)vt; a.Jpad = Jpad; a.out = (bf16*)out; a.lse = lse;
    dim3 grid((d->T + 63) / 64, d->H, d->B);
    if (d->DH == 64) COMMU_LAUNCH(relattn_fwd_kernel<64>, grid, dim3(256), 0, stream, a);
    else if (d->DH == 32) COMMU_LAUNCH(relattn_fwd_kernel<32>, grid, dim3(256), 0, stream, a);
    else return -22;
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_relattn_bwd(const commu_attn_desc* d, const commu_attn_bwd_desc* e, hipStream_t stream) {
    if (d->T <= 0 || d->B <= 0) return 0;
    const int K = d->T + d->M;
    if ((d->ld_qkv % 8) || (d->ld_rd % 8) || (e->Jpad % 64) || (e->Tpad % 64) || (e->Wr % 8) ||
        e->Jpad < ((K + 63) / 64) * 64 || e->Tpad < ((d->T + 63) / 64) * 64 || e->Wr < 128 + 8 + K + 192)
        return -22;
    AttnArgs a = {};
    fill_common(a, d);
    a.o = (const bf16*)e->o; a.dout = (const bf16*)e->dout; a.lse_in = e->lse; a.delta = e->delta;
    a.kt = (const bf16*)e->kt; a.rdt = (const bf16*)e->rdt; a.qut = (const bf16*)e->qut; a.dot = (const bf16*)e->dot;
    a.dq = (bf16*)e->dq; a.dk = (bf16*)e->dk; a.dv = (bf16*)e->dv; a.qv_out = (bf16*)e->qv_out;
    a.dsk = (bf16*)e->dsk; a.du_part = e->du_part; a.dvb_part = e->dvb_part;
    a.ld_dqkv = e->ld_dqkv; a.ld_dsk = e->ld_dsk; a.Jpad = e->Jpad; a.Tpad = e->Tpad; a.Wr = e->Wr;
    a.sft = (8 - ((d->M + 1) % 8)) % 8;
    dim3 gq((d->T + 63) / 64, d->H, d->B), gk((K + 63) / 64, d->H, d->B);
    if (d->DH == 64) {
        COMMU_LAUNCH(relattn_bwd_q_kernel<64>, gq, dim3(256), 0, stream, a);
        COMMU_LAUNCH(relattn_bwd_kv_kernel<64>, gk, dim3(256), 0, stream, a);
    } else if (d->DH == 32) {
        COMMU_LAUNCH(relattn_bwd_q_kernel<32>, gq, dim3(256), 0, stream, a);
        COMMU_LAUNCH(relattn_bwd_kv_kernel<32>, gk, dim3(256), 0, stream, a);
    } else return -22;
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_attn_rdt_shift(int M) { return (8 - ((M + 1) % 8)) % 8; }

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
