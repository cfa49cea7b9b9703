// fp32 PARITY MODE of the generation path (model.parity_fp32 / generate.py --parity).
//
// The reference computes everything in fp32 (train.py:48 `amp = None`; no autocast anywhere in commu/model/model.py), and
// BASELINE.json's north star asks for bit-exact greedy tokens.  The throughput path of this library multiplies bf16
// operands, so its greedy argmax agrees with the reference only where the top-1 / top-2 logit gap exceeds the bf16 error.
// This translation unit is the same forward math -- embedding, sinusoid table, Linear, relative-position attention with XL
// memory, LayerNorm, logits (model.py:64-73,142-147,163-181,283-352,409-420,578-626) -- on fp32 operands end to end:
// fp32 master weights (no shadows), fp32 activations, fp32 K/V cache, fp32 MFMA (v_mfma_f32_16x16x4_f32: full-precision
// products, fp32 accumulation) for the Linears, accurate expf / sinf / cosf.  What differs from the reference's CPU / GPU
// PyTorch kernels is summation ORDER only (logits agree to ~1e-6 of their range).
// The decode step is HBM-bound either way; fp32 costs 2x the bytes of the bf16 path.
#include "common.h"
#include "commu_hip.h"
#include <math.h>

typedef __attribute__((ext_vector_type(4))) float f4;

// ------------------------------------------------------------------------------------------------ Linear
// C[M,N] = A[M,K] . B[N,K]^T (+ bias[n]) (ReLU) (+ resid[m,n]); every operand fp32, row-major with leading dimensions.
// 64 x 64 tile, 4 waves (each 32 x 32 = 2 x 2 MFMA tiles of 16 x 16), K step 16.
// v_mfma_f32_16x16x4_f32: A lane l = A[i = l & 15][k = l >> 4], B lane l = B[k = l >> 4][j = l & 15],
// D lane l, register r = D[i = 4 (l >> 4) + r][j = l & 15].
#define PG_BM 64
#define PG_BN 64
#define PG_BK 16
#define PG_LD 80          // LDS row pitch in floats: k rows 0 / 1 of a 32-lane read land in banks 0-15 / 16-31

template <bool VEC>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                          int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                          const float* __restrict__ bias, const float* __restrict__ resid,
                                                          int ldr, int relu) {
    __shared__ float As[PG_BK][PG_LD];
    __shared__ float Bs[PG_BK][PG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * PG_BM, n0 = blockIdx.x * PG_BN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lr = tid >> 2, lk = (tid & 3) * 4;          // this thread stages row lr, k columns lk .. lk + 3 of both tiles
    const int am = m0 + lr, bn = n0 + lr;
    const float* ap = A + (size_t)(am < M ? am : 0) * lda;
    const float* bp = B + (size_t)(bn < N ? bn : 0) * ldb;
    for (int k0 = 0; k0 < K; k0 += PG_BK) {
        float av[4], bv[4];
        const int kk = k0 + lk;
        if (VEC && kk + 3 < K) {
            const f4 x = *(const f4*)(ap + kk), y = *(const f4*)(bp + kk);
            av[0] = x.x; av[1] = x.y; av[2] = x.z; av[3] = x.w;
            bv[0] = y.x; bv[1] = y.y; bv[2] = y.z; bv[3] = y.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[e] = (kk + e < K) ? ap[kk + e] : 0.f;
                bv[e] = (kk + e < K) ? bp[kk + e] : 0.f;
            }
        }
        if (am >= M) av[0] = av[1] = av[2] = av[3] = 0.f;
        if (bn >= N) bv[0] = bv[1] = bv[2] = bv[3] = 0.f;
        __syncthreads();          // the previous step's fragment reads are done
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[lk + e][lr] = av[e];
            Bs[lk + e][lr] = bv[e];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < PG_BK; ks += 4) {
            const int kr = ks + (lane >> 4), c = lane & 15;
            const float a0 = As[kr][wm + c], a1 = As[kr][wm + 16 + c];
            const float b0 = Bs[kr][wn + c], b1 = Bs[kr][wn + 16 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wn + 16 * b + (lane & 15);
            if (col >= N) continue;
            const float bs = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + 16 * a + 4 * (lane >> 4) + r;
                if (row >= M) continue;
                float v = acc[a][b][r] + bs;
                if (relu) v = fmaxf(v, 0.f);
                if (resid) v += resid[(size_t)row * ldr + col];
                C[(size_t)row * ldc + col] = v;
            }
        }
}

// Decode-step form (M <= 64 rows: one token per sequence): a workgroup owns 16 output columns and all rows, its four waves
// split K four ways and every lane feeds FOUR MFMAs from one 16-byte load per operand -- lane group g of k-block t holds
// k = 16 t + 4 g + s for MFMA s, the same bijection on both operands --, partial tiles are added through LDS.  N / 16
// workgroups stream the weight matrix once (the 64 x 64 tile above would put 24 workgroups on a [1536, 512] weight).
__global__ __launch_bounds__(256) void gemm_nt_f32_skinny_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                                 int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                                 const float* __restrict__ bias, const float* __restrict__ resid,
                                                                 int ldr, int relu) {
    __shared__ f32x4 part[4][4][64];          // [wave][row tile][lane]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int ncol = min(n0 + r16, N - 1);
    const float* bp = B + (size_t)ncol * ldb;
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kq = (K + 63) / 64 * 16;          // k per wave, a multiple of 16
    const int kbeg = w * kq, kend = min(K, kbeg + kq);
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        const int kk = k0 + 4 * g;
        const f4 z = {0.f, 0.f, 0.f, 0.f};
        const f4 bv = (kk + 3 < kend) ? *(const f4*)(bp + kk) : z;          // (K % 4 == 0, 16-byte aligned rows: the launcher checks)
        f4 av[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int row = 16 * mt + r16;
            av[mt] = (row < M && kk + 3 < kend) ? *(const f4*)(A + (size_t)row * lda + kk) : z;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt].x, bv.x, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt].y, bv.y, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt].z, bv.z, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt].w, bv.w, acc[mt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) part[w][mt][lane] = acc[mt];
    __syncthreads();
    // wave w finishes row tile w: D lane l, register r = C[16 w + 4 (l >> 4) + r][n0 + (l & 15)]
    f32x4 sum = part[0][w][lane];
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) sum += part[ww][w][lane];
    const int col = n0 + r16;
    if (col >= N) return;
    const float bs = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + 4 * g + r;
        if (row >= M) continue;
        float v = sum[r] + bs;
        if (relu) v = fmaxf(v, 0.f);
        if (resid) v += resid[(size_t)row * ldr + col];
        C[(size_t)row * ldc + col] = v;
    }
}

extern "C" int commu_gemm_nt_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                                 const float* bias, const float* resid, int ldr, int relu, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return -22;
    if (M <= 64 && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && (((uintptr_t)A | (uintptr_t)B) % 16 == 0)) {
        COMMU_LAUNCH(gemm_nt_f32_skinny_kernel, dim3((N + 15) / 16), dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, bias,
                     resid, ldr, relu);
        COMMU_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid((N + PG_BN - 1) / PG_BN, (M + PG_BM - 1) / PG_BM);
    const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A | (uintptr_t)B) % 16 == 0);
    if (vec)
        COMMU_LAUNCH(gemm_nt_f32_kernel<true>, grid, dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, bias, resid, ldr, relu);
    else
        COMMU_LAUNCH(gemm_nt_f32_kernel<false>, grid, dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, bias, resid, ldr, relu);
    COMMU_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ embedding, sinusoid, LayerNorm
// model.py:409-420: out[row] = E[token[row]] * sqrt(d_model)
__global__ void embed_f32_kernel(const long long* __restrict__ tok, const float* __restrict__ E, float* __restrict__ out, int ld,
                                 int rows, int D, float scale) {
    const int row = blockIdx.x;
    const float* e = E + (size_t)tok[row] * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) out[(size_t)row * ld + d] = e[d] * scale;
}

extern "C" int commu_embed_f32(const long long* tok, const float* E, float* out, int ld, int rows, int D, float scale,
                               hipStream_t stream) {
    if (rows <= 0) return -22;
    COMMU_LAUNCH(embed_f32_kernel, dim3(rows), dim3(128), 0, stream, tok, E, out, ld, rows, D, scale);
    COMMU_LAUNCH_CHECK();
    return 0;
}

// model.py:142-147 by DISTANCE: out[d] = [sin(d * inv_freq) | cos(d * inv_freq)], d = 0 .. n - 1 (the reference's row m of a
// K-row table is position K - 1 - m, and the rel-shift pairs query i / key j with position i + M - j: the distance)
__global__ void posemb_f32_kernel(const float* __restrict__ inv_freq, float* __restrict__ out, int ld, int n, int D, int clamp_len) {
    const int d = blockIdx.x;
    const int half = D / 2;
    for (int c = threadIdx.x; c < half; c += blockDim.x) {
        const float x = (float)(clamp_len > 0 ? min(d, clamp_len) : d) * inv_freq[c];          // (torch.ger: one fp32 product per entry; :581-582)
        out[(size_t)d * ld + c] = sinf(x);
        out[(size_t)d * ld + half + c] = cosf(x);
    }
}

extern "C" int commu_posemb_f32(const float* inv_freq, float* out, int ld, int n, int D, int clamp_len, hipStream_t stream) {
    if (n <= 0 || D % 2) return -22;
    COMMU_LAUNCH(posemb_f32_kernel, dim3(n), dim3(128), 0, stream, inv_freq, out, ld, n, D, clamp_len);
    COMMU_LAUNCH_CHECK();
    return 0;
}

// nn.LayerNorm (model.py:179,352): biased variance, eps inside the root; one wave per row, two passes over registers
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                                            const float* __restrict__ b, float* __restrict__ y, int ldy,
                                                            int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float v[16];          // D <= 1024
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int c = lane + 64 * e;
        v[e] = c < D ? xr[c] : 0.f;
        s += v[e];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int c = lane + 64 * e;
        const float t = c < D ? v[e] - mean : 0.f;
        q += t * t;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int c = lane + 64 * e;
        if (c < D) y[(size_t)row * ldy + c] = (v[e] - mean) * rstd * g[c] + b[c];
    }
}

extern "C" int commu_layernorm_f32(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, int rows,
                                   int D, float eps, hipStream_t stream) {
    if (rows <= 0 || D <= 0 || D > 1024) return -22;
    COMMU_LAUNCH(layernorm_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, ldx, gamma, beta, y, ldy, rows, D, eps);
    COMMU_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ attention
// model.py:283-345 for one (query i, sequence b, head h) per wave:
//   s_j = ((q_i + u) . k_j + (q_i + v) . Rd[i + M_b - j]) * scale,  visible keys: lo_i <= j <= i + M_b,
//   out_i = softmax_j(s) . v.
// Keys in chunks of 64 (a lane owns a key: its dot products run over d in order), online softmax across chunks, the
// probability-weighted sum of the value rows with a lane per feature.  Element (j, b, h, d) of k / v is at
// base + j * sj + b * sb + h * DH + d: the [K * B, 3 H DH] projection buffer (sj = B * ld, sb = ld) or the decode cache
// [B][Lmax][H DH] (sj = H DH, sb = Lmax * H DH).  klen (optional, int32 [B]): the memory length M_b of sequence b (ragged
// decode batch); otherwise M for every sequence.  Masks as model.py:549-574: causal; same_length (keys j <= i - s hidden, s
// from the sequence's own key count); reset[b] (memory keys hidden).
template <int VW>
__global__ __launch_bounds__(64) void relattn_f32_kernel(const float* __restrict__ q, int ld_q, const float* __restrict__ kbase,
                                                         const float* __restrict__ vbase, long long sj, long long sb,
                                                         const float* __restrict__ rd, int ld_rd, const float* __restrict__ u,
                                                         const float* __restrict__ vbias, const int* __restrict__ klen,
                                                         const unsigned char* __restrict__ reset, float* __restrict__ out,
                                                         int ld_o, int T, int M, int B, int H, int DH, int same_length,
                                                         int mem_len, float scale) {
    __shared__ float qu[64], qv[64];
    const int h = blockIdx.x, b = blockIdx.y, i = blockIdx.z, lane = threadIdx.x;
    const int Mb = klen ? klen[b] : M;
    const float* qr = q + ((size_t)i * B + b) * ld_q + (size_t)h * DH;
    if (lane < DH) {
        const float x = qr[lane];
        qu[lane] = x + u[h * DH + lane];
        qv[lane] = x + vbias[h * DH + lane];
    }
    __syncthreads();
    const int Kb = Mb + T;
    int lo = 0;
    if (same_length) {
        const int mask_len = Kb - mem_len;
        const int s = mask_len > 0 ? T - mask_len : T;
        lo = i - s + 1;          // keys j <= i - s are hidden
        if (lo < 0) lo = 0;
    }
    if (reset && reset[b] && lo < Mb) lo = Mb;
    const int hi = i + Mb;      // last visible key
    const float* kb = kbase + (size_t)b * sb + (size_t)h * DH;
    const float* vb_ = vbase + (size_t)b * sb + (size_t)h * DH;
    const float* rdh = rd + (size_t)h * DH;
    float mrun = -INFINITY, lrun = 0.f, acc = 0.f;
    for (int j0 = lo; j0 <= hi; j0 += 64) {
        const int j = j0 + lane;
        float s = -INFINITY;
        if (j <= hi) {
            const float* kr = kb + (size_t)j * sj;
            const float* rr = rdh + (size_t)(i + Mb - j) * ld_rd;
            float ac = 0.f, bd = 0.f;
            if (VW == 4) {
                for (int d = 0; d < DH; d += 4) {
                    const f4 kx = *(const f4*)(kr + d), rx = *(const f4*)(rr + d);
                    ac = fmaf(qu[d], kx.x, ac); ac = fmaf(qu[d + 1], kx.y, ac); ac = fmaf(qu[d + 2], kx.z, ac); ac = fmaf(qu[d + 3], kx.w, ac);
                    bd = fmaf(qv[d], rx.x, bd); bd = fmaf(qv[d + 1], rx.y, bd); bd = fmaf(qv[d + 2], rx.z, bd); bd = fmaf(qv[d + 3], rx.w, bd);
                }
            } else {
                for (int d = 0; d < DH; ++d) {
                    ac = fmaf(qu[d], kr[d], ac);
                    bd = fmaf(qv[d], rr[d], bd);
                }
            }
            s = (ac + bd) * scale;
        }
        const float mnew = fmaxf(mrun, wave_max(s));
        const float p = (j <= hi) ? expf(s - mnew) : 0.f;
        const float corr = (mrun == -INFINITY) ? 0.f : expf(mrun - mnew);
        lrun = lrun * corr + wave_sum(p);
        acc *= corr;
        const int n = min(64, hi - j0 + 1);
        // (v_readlane reads the probability of key jj whatever the EXEC mask: with d_head < 64 the lanes that own the
        //  keys d_head .. 63 of a chunk are not among the feature lanes below)
        const int pbits = __builtin_bit_cast(int, p);
        if (lane < DH) {
            const float* vr = vb_ + (size_t)j0 * sj + lane;
            for (int jj = 0; jj < n; ++jj)
                acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(pbits, jj)), vr[(size_t)jj * sj], acc);
        }
        mrun = mnew;
    }
    if (lane < DH) out[((size_t)i * B + b) * ld_o + (size_t)h * DH + lane] = (hi >= lo) ? acc / lrun : 0.f;
}

extern "C" int commu_relattn_f32(const float* q, int ld_q, const float* k, const float* v, long long stride_key,
                                 long long stride_seq, const float* rd, int ld_rd, const float* r_w_bias, const float* r_r_bias,
                                 const int* klen, const unsigned char* reset, float* out, int ld_o, int T, int M, int B, int H,
                                 int DH, int same_length, int mem_len, float scale, hipStream_t stream) {
    if (T <= 0 || B <= 0 || H <= 0 || DH <= 0 || DH > 64 || B > 65535 || T > 65535) return -22;
    const dim3 grid(H, B, T);
    const bool v4 = DH % 4 == 0 && stride_key % 4 == 0 && stride_seq % 4 == 0 && ld_rd % 4 == 0 &&
                    (((uintptr_t)k | (uintptr_t)rd) % 16 == 0);
    if (v4)
        COMMU_LAUNCH(relattn_f32_kernel<4>, grid, dim3(64), 0, stream, q, ld_q, k, v, stride_key, stride_seq, rd, ld_rd, r_w_bias,
                     r_r_bias, klen, reset, out, ld_o, T, M, B, H, DH, same_length, mem_len, scale);
    else
        COMMU_LAUNCH(relattn_f32_kernel<1>, grid, dim3(64), 0, stream, q, ld_q, k, v, stride_key, stride_seq, rd, ld_rd, r_w_bias,
                     r_r_bias, klen, reset, out, ld_o, T, M, B, H, DH, same_length, mem_len, scale);
    COMMU_LAUNCH_CHECK();
    return 0;
}

// decode step: the new token's key / value rows (columns HD .. 3 HD of its projection) into row klen[b] of the caches
// [B][Lmax][HD] for the sequences that step (active == NULL: all)
__global__ void kv_append_f32_kernel(const float* __restrict__ qkv, int ld, float* __restrict__ kc, float* __restrict__ vc,
                                     const int* __restrict__ klen, const unsigned char* __restrict__ active, int HD, int Lmax) {
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const int pos = klen[b];
    if (pos >= Lmax) return;
    const float* src = qkv + (size_t)b * ld;
    float* kd = kc + ((size_t)b * Lmax + pos) * HD;
    float* vd = vc + ((size_t)b * Lmax + pos) * HD;
    for (int c = threadIdx.x; c < HD; c += blockDim.x) {
        kd[c] = src[HD + c];
        vd[c] = src[2 * HD + c];
    }
}

extern "C" int commu_decode_kv_append_f32(const float* qkv, int ld, float* kc, float* vc, const int* klen,
                                          const unsigned char* active, int B, int HD, int Lmax, hipStream_t stream) {
    if (B <= 0) return -22;
    COMMU_LAUNCH(kv_append_f32_kernel, dim3(B), dim3(128), 0, stream, qkv, ld, kc, vc, klen, active, HD, Lmax);
    COMMU_LAUNCH_CHECK();
    return 0;
}
