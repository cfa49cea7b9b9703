// Decode step, everything of a layer that follows its attention, as ONE launch (gfx950).
//
// Reference: the qlen = 1 forward_generate step (commu/model/model.py:606-628) runs, per layer, o_net + residual +
// LayerNorm (model.py:344-352), the position-wise FFN + residual + LayerNorm (model.py:163-179) and then the next
// layer's qkv_net (model.py:297-299) -- or, after the last layer, the tied-embedding logits (model.py:64-73).  For 64
// sequences those are four Linear layers over a [64, 512] activation: 4.2 MB of weights, micro-seconds of arithmetic.  As
// separate launches (csrc/gemm.hip skinny kernel + layernorm_fwd) each costs 4-10 us of launch, first-byte latency and
// drain: 36 us per layer, 75 % of the decode iteration.
//
// Here the four Linears are four PHASES of one launch.  Workgroup (mg, ng) owns 16 sequences (row group mg) and, in every
// phase, the output-column tiles ng, ng + 32, ... of that phase (16 columns each):
//   phase 1  z1  = vec . Wo^T + h                                  [16 x D]   K = HD
//   phase 2  a   = LN1(z1);  hid = relu(a . W1^T + b1)             [16 x DI]  K = D
//   phase 3  z2  = hid . W2^T + b2 + a                             [16 x D]   K = DI
//   phase 4  h'  = LN2(z2);  out = h' . Wn^T (+ bn)                [16 x Nn]  K = D     (next qkv_net, or the logits)
// A phase needs whole rows of the previous phase's output, which the 32 workgroups of the row group produced: between
// phases they meet at an arrival counter (one per phase and row group).  Data crosses workgroups the way the CDNA4 guide
// prescribes for hand-offs inside a launch, independent of where the workgroups run: the producer stores WRITE-THROUGH
// (sc1), every storing wave drains its stores (s_waitcnt vmcnt(0)), one lane adds to the counter with an agent-scope
// atomic; the consumer polls that one word with relaxed agent-scope loads from one lane, then every wave reads the
// payload with sc1 loads (which bypass the CU's L1).  Each hand-off buffer is written once per launch and belongs to
// one layer.  Spins are bounded: a workgroup that gives up sets *err and carries on, so a launch always terminates.
//
// Operands go straight from global memory into MFMA fragments (as in the skinny kernel): the four waves split K, the
// partial tiles are added through LDS.  The weights, biases and LayerNorm parameters of phase p + 1 are requested before
// the wait that ends phase p, so that the wait hides their latency.
#include "common.cuh"
#include "commu_hip.h"
#include <string.h>

namespace {

typedef __amdgpu_buffer_rsrc_t srd_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ srd_t make_srd(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(unsigned)bytes, 0x00020000);
}
constexpr int AUX_SC1 = 16;          // cache-policy bit of the raw buffer builtins: sc1 (agent scope: L1 bypass / write-through)
constexpr int NGRP = 32;             // column groups (= workgroups) per row group
constexpr unsigned SPIN_LIMIT = 1u << 17;

struct TailArgs {
    const bf16* vec;  int ld_vec;           // [B][HD] attention output
    const bf16* h;    int ld_h;             // [B][D] layer input (residual of o_net)
    const bf16* Wo;   int ld_wo;            // [D][HD]
    const bf16* W1;   int ld_w1;            // [DI][D]
    const bf16* W2;   int ld_w2;            // [D][DI]
    const bf16* Wn;   int ld_wn;  int Nn;   // [Nn][D]: next layer's qkv_net, or the embedding (logits)
    const float *b1, *b2, *bn;
    const float *g1, *be1, *g2, *be2;
    float eps1, eps2;
    int d_ln;                               // LayerNorm width (< D for zero-padded models)
    bf16 *z1, *hid, *z2;                    // hand-off buffers of this layer: [B][D], [B][DI], [B][D], dense
    bf16* h_out;      int ld_ho;            // [B][D] layer output
    void* out_n;      int ld_on;            // [B][Nn]: bf16 (qkv) or fp32 (logits)
    int B;
    unsigned* sync;                         // [3][4] arrival counters of this launch, zero on entry
    unsigned* err;
    const unsigned char* active;            // (logits) rows with active[row] == 0 keep their previous logits; null: all
    // head launch (MODE_HEAD): x = E32[tok[row]] * emb_scale instead of LN2(z2); zero_words[0 .. n_zero) are cleared
    const int64_t* tok;
    const float* E32;
    int d_true, V;
    float emb_scale;
    unsigned* zero_words;
    int n_zero;
};
enum { MODE_QKV = 0, MODE_LOGITS = 1, MODE_HEAD = 2 };

template <int KS>
__device__ __forceinline__ void load_x(bf16x8 (&xf)[KS], srd_t srd, unsigned off, bool sc1) {
    // (two code paths so that the cache policy is an immediate)
    if (sc1) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(off + 64u * ks), 0, AUX_SC1));
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(off + 64u * ks), 0, 0));
    }
}

// weight fragments: tile t of this workgroup is output columns 16 (ng + 32 t) .. + 16; lane (r16, g) holds
// W[16 tile + r16][k0 + 32 ks + 8 g .. + 8]  (rows past the matrix are clamped: their products are never stored)
template <int KS, int NT>
__device__ __forceinline__ void load_w(bf16x8 (&wf)[NT][KS], const bf16* __restrict__ W, int ldw, int nrows, int ng,
                                       int r16, int k0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bf16* wp = W + (size_t)min(16 * (ng + NGRP * t) + r16, nrows - 1) * ldw + k0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[t][ks] = ld_bf16x8(wp + 32 * ks);
    }
}

template <int KS>
__device__ __forceinline__ void load_affine(float (&gm)[KS][8], float (&bt)[KS][8], const float* __restrict__ gamma,
                                            const float* __restrict__ beta, int d_ln, int k0) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = k0 + 32 * ks + e;
            gm[ks][e] = kk < d_ln ? gamma[kk] : 0.f;
            bt[ks][e] = kk < d_ln ? beta[kk] : 0.f;
        }
}

// LayerNorm of the 16 rows held as fragments (two passes: mean, centred squares -- like layernorm_fwd_kernel)
template <int KS>
__device__ __forceinline__ void layer_norm(bf16x8 (&xf)[KS], const float (&gm)[KS][8], const float (&bt)[KS][8], int d_ln,
                                           float eps, float (*st)[16], int w, int r16, int g, int k0) {
    float mu = 0.f, rs = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float v = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool in = k0 + 32 * ks + e < d_ln;
                const float x = in ? bf2f(xf[ks][e]) : 0.f;
                v += pass == 0 ? x : (in ? (x - mu) * (x - mu) : 0.f);
            }
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) st[w][r16] = v;
        __syncthreads();
        const float t = st[0][r16] + st[1][r16] + st[2][r16] + st[3][r16];
        if (pass == 0) mu = t / (float)d_ln;
        else rs = rsqrtf(t / (float)d_ln + eps);
        __syncthreads();
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            xf[ks][e] = f2bf(k0 + 32 * ks + e < d_ln ? (bf2f(xf[ks][e]) - mu) * rs * gm[ks][e] + bt[ks][e] : 0.f);
}

template <int KS, int NT>
__device__ __forceinline__ void multiply(const bf16x8 (&wf)[NT][KS], const bf16x8 (&xf)[KS], float (*red)[12][64], int w,
                                         int lane) {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma16(wf[t][ks], xf[ks], acc[t]);
    // lane (r16, g) now holds C[row r16][col 16 tile + 4 g + e] of this wave's K range
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[w][4 * t + e][lane] = acc[t][e];
    __syncthreads();
}

__device__ __forceinline__ f32x4 tile_sum(float (*red)[12][64], int t, int lane) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = red[0][4 * t + e][lane] + red[1][4 * t + e][lane] + red[2][4 * t + e][lane] + red[3][4 * t + e][lane];
    return v;
}

// end of a producing phase: every wave drains its write-through stores, then one lane counts the workgroup in
__device__ __forceinline__ void arrive(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void wait_arrivals(unsigned* cnt, unsigned* err, unsigned code) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NGRP) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) {
                __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void store4_sc1(srd_t srd, unsigned off, f32x4 v) {
    const bf16x4 o = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), srd, (int)off, 0, AUX_SC1);
}

template <int D, int DI, int HD, int MODE>
__global__ __launch_bounds__(256) void decode_tail_kernel(TailArgs a) {
    constexpr bool LOGITS = MODE == MODE_LOGITS, HEAD = MODE == MODE_HEAD;
    constexpr int KS_D = D / 128, KS_DI = DI / 128, KS_HD = HD / 128;        // 32-wide MFMA steps per wave (4 waves split K)
    constexpr int NT1 = D / 512, NT2 = DI / 512, NT3 = D / 512, NT4 = LOGITS ? 2 : (3 * HD) / 512;
    static_assert(D % 512 == 0 && DI % 512 == 0 && HD % 512 == 0 && NT2 <= 3 && NT4 <= 3 && NT1 == 1, "shape");
    __shared__ float red[4][12][64];
    __shared__ float st[4][16];
    __shared__ __attribute__((aligned(16))) bf16 abuf[16][16];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int ng = blockIdx.x & (NGRP - 1), mg = blockIdx.x >> 5;
    const int B = a.B, row = 16 * mg + r16;
    unsigned* cnt = a.sync + mg;
    // the lanes that hold columns 16 ng .. 16 ng + 16 of a [16 x D] row block whose K range the waves split: wave, step, g pair
    const int own_w = (16 * ng) / (32 * KS_D), own_ks = ((16 * ng) % (32 * KS_D)) / 32, own_g2 = ((16 * ng) % 32) / 16;
    const bool own = (w == own_w) && ((g >> 1) == own_g2);

    if (HEAD && blockIdx.x == 0)
        for (int i = tid; i < a.n_zero; i += 256) a.zero_words[i] = 0u;
    // ---------------------------------------------------------------- phase 1: z1 = vec . Wo^T + h
    if (!HEAD) {
        const int k0 = w * (32 * KS_HD) + 8 * g;
        bf16x8 wf[NT1][KS_HD], xf[KS_HD];
        load_w<KS_HD, NT1>(wf, a.Wo, a.ld_wo, D, ng, r16, k0);
        const srd_t sx = make_srd(a.vec, ((size_t)(B - 1) * a.ld_vec + HD) * 2);
        load_x<KS_HD>(xf, sx, ((unsigned)row * a.ld_vec + k0) * 2u, false);
        bf16x4 res = (bf16x4){f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
        if (w < NT1 && row < B) res = *(const bf16x4*)(a.h + (size_t)row * a.ld_h + 16 * (ng + NGRP * w) + 4 * g);
        multiply<KS_HD, NT1>(wf, xf, red, w, lane);
        if (w < NT1 && row < B) {
            f32x4 v = tile_sum(red, w, lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bf2f(res[e]);
            store4_sc1(make_srd(a.z1, (size_t)B * D * 2), ((unsigned)row * D + 16 * (ng + NGRP * w) + 4 * g) * 2u, v);
        }
        arrive(cnt + 0);
    }
    // ---------------------------------------------------------------- phase 2: a = LN1(z1); hid = relu(a . W1^T + b1)
    if (!HEAD) {
        const int k0 = w * (32 * KS_D) + 8 * g;
        bf16x8 wf[NT2][KS_D], xf[KS_D];
        float gm[KS_D][8], bt[KS_D][8];
        load_w<KS_D, NT2>(wf, a.W1, a.ld_w1, DI, ng, r16, k0);
        load_affine<KS_D>(gm, bt, a.g1, a.be1, a.d_ln, k0);
        f32x4 bias = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (w < NT2) bias = *(const f32x4*)(a.b1 + 16 * (ng + NGRP * w) + 4 * g);
        wait_arrivals(cnt + 0, a.err, 1u);
        load_x<KS_D>(xf, make_srd(a.z1, (size_t)B * D * 2), ((unsigned)row * D + k0) * 2u, true);
        layer_norm<KS_D>(xf, gm, bt, a.d_ln, a.eps1, st, w, r16, g, k0);
        if (own) {
#pragma unroll
            for (int ks = 0; ks < KS_D; ++ks)
                if (ks == own_ks) *(bf16x8*)&abuf[r16][8 * (g & 1)] = xf[ks];
        }
        multiply<KS_D, NT2>(wf, xf, red, w, lane);
        if (w < NT2 && row < B) {
            f32x4 v = tile_sum(red, w, lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bias[e], 0.f);
            store4_sc1(make_srd(a.hid, (size_t)B * DI * 2), ((unsigned)row * DI + 16 * (ng + NGRP * w) + 4 * g) * 2u, v);
        }
        arrive(cnt + 4);
    }
    // ---------------------------------------------------------------- phase 3: z2 = hid . W2^T + b2 + a
    if (!HEAD) {
        const int k0 = w * (32 * KS_DI) + 8 * g;
        bf16x8 wf[NT3][KS_DI], xf[KS_DI];
        load_w<KS_DI, NT3>(wf, a.W2, a.ld_w2, D, ng, r16, k0);
        f32x4 bias = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (w < NT3) bias = *(const f32x4*)(a.b2 + 16 * (ng + NGRP * w) + 4 * g);
        wait_arrivals(cnt + 4, a.err, 2u);
        load_x<KS_DI>(xf, make_srd(a.hid, (size_t)B * DI * 2), ((unsigned)row * DI + k0) * 2u, true);
        multiply<KS_DI, NT3>(wf, xf, red, w, lane);
        if (w < NT3 && row < B) {
            f32x4 v = tile_sum(red, w, lane);
            const bf16x4 res = *(const bf16x4*)&abuf[r16][4 * g];          // (NT3 == 1: the tile is columns 16 ng .. + 16)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bias[e] + bf2f(res[e]);
            store4_sc1(make_srd(a.z2, (size_t)B * D * 2), ((unsigned)row * D + 16 * (ng + NGRP * w) + 4 * g) * 2u, v);
        }
        arrive(cnt + 8);
    }
    // ---------------------------------------------------------------- phase 4: h' = LN2(z2); out = h' . Wn^T (+ bn)
    {
        const int k0 = w * (32 * KS_D) + 8 * g;
        bf16x8 wf[NT4][KS_D], xf[KS_D];
        float gm[KS_D][8], bt[KS_D][8];
        load_w<KS_D, NT4>(wf, a.Wn, a.ld_wn, a.Nn, ng, r16, k0);
        if (!HEAD) load_affine<KS_D>(gm, bt, a.g2, a.be2, a.d_ln, k0);
        const int col = 16 * (ng + NGRP * w) + 4 * g;
        f32x4 bias = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (LOGITS && w < NT4 && a.bn != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[e] = col + e < a.Nn ? a.bn[col + e] : 0.f;
        }
        if (HEAD) {
            // x = E[tok[row]] * sqrt(d_model) (commu/model/model.py:409-420); an id outside the vocabulary poisons its row
            // with NaN, like commu_embed_fwd
            const long long id = row < B ? a.tok[row] : 0;
            const bool bad = id < 0 || id >= a.V;
            const float* src = a.E32 + (size_t)(bad ? 0 : id) * a.d_true;
#pragma unroll
            for (int ks = 0; ks < KS_D; ++ks) {
                const int kk = k0 + 32 * ks;
                f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
                if (kk < a.d_true) v0 = *(const f32x4*)(src + kk);
                if (kk + 4 < a.d_true) v1 = *(const f32x4*)(src + kk + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = bad ? __builtin_nanf("") : (e < 4 ? v0[e & 3] : v1[e & 3]) * a.emb_scale;
                    xf[ks][e] = f2bf(kk + e < a.d_true ? x : 0.f);
                }
            }
        } else {
            wait_arrivals(cnt + 8, a.err, 3u);
            load_x<KS_D>(xf, make_srd(a.z2, (size_t)B * D * 2), ((unsigned)row * D + k0) * 2u, true);
            layer_norm<KS_D>(xf, gm, bt, a.d_ln, a.eps2, st, w, r16, g, k0);
        }
        if (own && row < B && a.h_out != nullptr) {
#pragma unroll
            for (int ks = 0; ks < KS_D; ++ks)
                if (ks == own_ks) st_bf16x8(a.h_out + (size_t)row * a.ld_ho + 16 * ng + 8 * (g & 1), xf[ks]);
        }
        multiply<KS_D, NT4>(wf, xf, red, w, lane);
        if (w < NT4 && row < B && col < a.Nn && !(LOGITS && a.active != nullptr && !a.active[row])) {
            f32x4 v = tile_sum(red, w, lane);
            if (LOGITS) {
                float* o = (float*)a.out_n + (size_t)row * a.ld_on + col;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < a.Nn) o[e] = v[e] + bias[e];
            } else {
                bf16* o = (bf16*)a.out_n + (size_t)row * a.ld_on + col;
                *(bf16x4*)o = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            }
        }
    }
}

}  // namespace

extern "C" int commu_decode_tail_supported(int B, int D, int DI, int HD) {
    return (B >= 1 && B <= 64 && D == 512 && DI == 1024 && HD == 512) ? 1 : 0;
}

extern "C" int commu_decode_tail_sync_words(void) { return 12; }

static TailArgs tail_args_zero() {
    TailArgs a;
    memset(&a, 0, sizeof(a));
    return a;
}

extern "C" int commu_decode_layer_tail(const void* vec, int ld_vec, const void* h, int ld_h, const void* Wo, int ld_wo,
                                       const void* W1, int ld_w1, const float* b1, const void* W2, int ld_w2,
                                       const float* b2, const float* g1, const float* be1, float eps1, const float* g2,
                                       const float* be2, float eps2, int d_ln, const void* Wn, int ld_wn, int Nn,
                                       const float* bn, int logits, const unsigned char* active, void* z1, void* hid,
                                       void* z2, void* h_out, int ld_ho, void* out_n, int ld_on, int B, int D, int DI,
                                       int HD, unsigned* sync, unsigned* err, hipStream_t stream) {
    if (!commu_decode_tail_supported(B, D, DI, HD)) return -22;
    if (d_ln <= 0 || d_ln > D || (ld_vec % 8) || (ld_h % 4) || (ld_wo % 8) || (ld_w1 % 8) || (ld_w2 % 8) || (ld_wn % 8) ||
        (ld_ho % 8) || (ld_on % 4) || sync == nullptr || err == nullptr)
        return -22;
    if (logits ? (Nn < 1 || Nn > 1024) : (Nn != 3 * HD)) return -22;
    TailArgs a = tail_args_zero();
    a.vec = (const bf16*)vec; a.ld_vec = ld_vec;
    a.h = (const bf16*)h; a.ld_h = ld_h;
    a.Wo = (const bf16*)Wo; a.ld_wo = ld_wo;
    a.W1 = (const bf16*)W1; a.ld_w1 = ld_w1;
    a.W2 = (const bf16*)W2; a.ld_w2 = ld_w2;
    a.Wn = (const bf16*)Wn; a.ld_wn = ld_wn; a.Nn = Nn;
    a.b1 = b1; a.b2 = b2; a.bn = bn;
    a.g1 = g1; a.be1 = be1; a.g2 = g2; a.be2 = be2;
    a.eps1 = eps1; a.eps2 = eps2; a.d_ln = d_ln;
    a.z1 = (bf16*)z1; a.hid = (bf16*)hid; a.z2 = (bf16*)z2;
    a.h_out = (bf16*)h_out; a.ld_ho = ld_ho;
    a.out_n = out_n; a.ld_on = ld_on;
    a.B = B; a.sync = sync; a.err = err; a.active = active;
    const dim3 grid(NGRP * ((B + 15) / 16));
    if (logits) COMMU_LAUNCH((decode_tail_kernel<512, 1024, 512, MODE_LOGITS>), grid, dim3(256), 0, stream, a);
    else COMMU_LAUNCH((decode_tail_kernel<512, 1024, 512, MODE_QKV>), grid, dim3(256), 0, stream, a);
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_decode_head(const int64_t* tok, const float* E, int d_true, int V, float scale, const void* Wqkv,
                                 int ld_w, void* h_out, int ld_ho, void* qkv, int ld_qkv, int B, int D, int DI, int HD,
                                 unsigned* zero_words, int n_zero, hipStream_t stream) {
    if (!commu_decode_tail_supported(B, D, DI, HD)) return -22;
    if (d_true <= 0 || d_true > D || (d_true % 4) || (ld_w % 8) || (ld_ho % 8) || (ld_qkv % 4) || h_out == nullptr) return -22;
    TailArgs a = tail_args_zero();
    a.tok = tok; a.E32 = E; a.d_true = d_true; a.V = V; a.emb_scale = scale;
    a.Wn = (const bf16*)Wqkv; a.ld_wn = ld_w; a.Nn = 3 * HD;
    a.h_out = (bf16*)h_out; a.ld_ho = ld_ho;
    a.out_n = qkv; a.ld_on = ld_qkv;
    a.B = B; a.zero_words = zero_words; a.n_zero = zero_words != nullptr ? n_zero : 0;
    const dim3 grid(NGRP * ((B + 15) / 16));
    COMMU_LAUNCH((decode_tail_kernel<512, 1024, 512, MODE_HEAD>), grid, dim3(256), 0, stream, a);
    COMMU_LAUNCH_CHECK();
    return 0;
}
