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
#include "common.h"
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
constexpr int CNT_STRIDE = 64;       // words between two arrival counters

struct TailArgs {
    const bf16* vec;  int ld_vec;           // [B][HD] attention output
    const bf16* h;    int ld_h;             // [B][D] layer input (residual of o_net)
    const bf16 *Wo, *W1, *W2, *Wn;          // PACKED (commu_decode_tail_pack) [D][HD], [DI][D], [D][DI], [Nn][D]
    int Nn;                                 // rows of Wn: next layer's qkv_net, or the embedding (logits)
    const float *b1, *b2, *bn;
    const float *g1, *be1, *g2, *be2;
    float eps1, eps2;
    int d_ln;                               // LayerNorm width (< D for zero-padded models)
    bf16 *z1, *hid, *z2;                    // hand-off buffers of this layer: [B][D], [B][DI], [B][D], dense
    bf16* h_out;      int ld_ho;            // [B][D] layer output
    void* out_n;      int ld_on;            // [B][Nn]: bf16 (qkv) or fp32 (logits)
    int B;
    unsigned* sync;                         // [3][4][CNT_STRIDE] arrival counters of this launch, zero on entry
    unsigned* err;
    const unsigned char* active;            // (logits) rows with active[row] == 0 keep their previous logits; null: all
    // head launch (MODE_HEAD): x = E32[tok[row]] * emb_scale instead of LN2(z2); zero_words[0 .. n_zero) are cleared
    const int64_t* tok;
    const float* E32;
    int d_true, V;
    float emb_scale;
    unsigned* zero_words;
    int n_zero;
    unsigned long long* trace;              // (diagnostics) [workgroup][16] 100 MHz timestamps of the phase boundaries, or null
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

// weight fragments from the PACKED copy (commu_decode_tail_pack): tile t of workgroup ng is output columns
// 16 (ng + 32 t) .. + 16; lane (r16, g) of wave w holds W[16 tile + r16][w 32 KS + 32 ks + 8 g .. + 8], stored at
// ((((ng NT + t) 4 + w) KS + ks) 64 + lane) x 16 bytes -- one wave instruction reads 1 KB of consecutive bytes
template <int KS, int NT>
__device__ __forceinline__ void load_w(bf16x8 (&wf)[NT][KS], const bf16* __restrict__ Wp, int ng, int w, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bf16* wp = Wp + ((size_t)(((ng * NT + t) * 4 + w) * KS) * 64 + lane) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[t][ks] = ld_bf16x8(wp + ks * 512);
    }
}

__global__ void pack_weight_kernel(const bf16* __restrict__ W, int ldw, int N, int KS, int NT, bf16* __restrict__ out,
                                   long long nchunks) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const int lane = (int)(c & 63);
    long long q = c >> 6;
    const int ks = (int)(q % KS); q /= KS;
    const int w = (int)(q & 3); q >>= 2;
    const int t = (int)(q % NT);
    const int ng = (int)(q / NT);
    const int r = 16 * (ng + NGRP * t) + (lane & 15), k = w * 32 * KS + 32 * ks + 8 * (lane >> 4);
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < N) v = ld_bf16x8(W + (size_t)r * ldw + k);
    st_bf16x8(out + c * 8, v);
}

// (unconditional 16-byte loads: gamma / beta hold at least d_ln floats, d_ln % 4 == 0; columns past d_ln are clamped here and
//  zeroed by layer_norm)
template <int KS>
__device__ __forceinline__ void load_affine(float (&gm)[KS][8], float (&bt)[KS][8], const float* __restrict__ gamma,
                                            const float* __restrict__ beta, int d_ln, int k0) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int h4 = 0; h4 < 2; ++h4) {
            const int kk = min(k0 + 32 * ks + 4 * h4, d_ln - 4);
            const f32x4 gv = *(const f32x4*)(gamma + kk), bv = *(const f32x4*)(beta + kk);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gm[ks][4 * h4 + e] = gv[e]; bt[ks][4 * h4 + e] = bv[e]; }
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
__device__ __forceinline__ void multiply(const bf16x8 (&wf)[NT][KS], const bf16x8 (&xf)[KS], float (*red)[16][64], int w,
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

__device__ __forceinline__ f32x4 tile_sum(float (*red)[16][64], int t, int lane) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = red[0][4 * t + e][lane] + red[1][4 * t + e][lane] + red[2][4 * t + e][lane] + red[3][4 * t + e][lane];
    return v;
}

// end of a producing phase: every wave that stored drains its write-through stores, then one lane counts the workgroup in
// (the other waves have loads of the next phase in flight: stores and loads share vmcnt, so they must not wait here)
__device__ __forceinline__ void arrive(unsigned* cnt, bool stored) {
    if (stored) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (polled by wave 3, which never stores before a wait: nothing but landed loads is ahead of the poll in its memory queue)
__device__ __forceinline__ void wait_arrivals(unsigned* cnt, unsigned* err, unsigned code) {
    if (threadIdx.x == 192) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NGRP) {
            __builtin_amdgcn_s_sleep(1);          // (the wait is insensitive to the poll rate: 0 .. 16 measured equal)
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
    constexpr int NT1 = D / 512, NT2 = DI / 512, NT3 = D / 512, NT4 = LOGITS ? 2 : (3 * HD + 511) / 512;
    static_assert(D % 512 == 0 && DI % 512 == 0 && HD % 128 == 0 && NT2 <= 4 && NT4 <= 4 && NT1 == 1, "shape");
    __shared__ float red[4][16][64];
    __shared__ float st[4][16];
    __shared__ __attribute__((aligned(16))) bf16 abuf[16][16];

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
    const int ng = blockIdx.x & (NGRP - 1), mg = blockIdx.x >> 5;
    const int B = a.B, row = 16 * mg + r16;
    unsigned* cnt = a.sync + mg * CNT_STRIDE;          // (one 256-byte line per counter: pollers of different row groups
                                                      //  and phases queue at different memory channels)
    // the lanes that hold columns 16 ng .. 16 ng + 16 of a [16 x D] row block whose K range the waves split: wave, step, g pair
    const int own_w = (16 * ng) / (32 * KS_D), own_ks = ((16 * ng) % (32 * KS_D)) / 32, own_g2 = ((16 * ng) % 32) / 16;
    const bool own = (w == own_w) && ((g >> 1) == own_g2);

    if (HEAD)
        for (int i = blockIdx.x * 256 + tid; i < a.n_zero; i += gridDim.x * 256) a.zero_words[i] = 0u;
    int tix = 0;
    auto stamp = [&]() {
        if (a.trace != nullptr && tid == 0) a.trace[blockIdx.x * 16 + tix] = wall_clock64();
        ++tix;
    };
    stamp();
    // ---- operand schedule.  The weights (+ LayerNorm parameters, biases) of phase p + 1 are requested during phase p: by
    // the waves that do not store in phase p right after their activation loads (a whole phase to arrive, and the
    // activations are not queued behind them), by the storing waves after their drain (stores and loads share vmcnt).
    // Every load is unconditional (clamped addresses) so that the compiler's vmcnt waits stay counted.
    const int k0h = w * (32 * KS_HD) + 8 * g, k0d = w * (32 * KS_D) + 8 * g, k0i = w * (32 * KS_DI) + 8 * g;
    const int col = 16 * (ng + NGRP * w) + 4 * g;          // first output column of the tile this wave finishes (tile w)
    const int rowc = min(row, B - 1);
    bf16x8 wf1[NT1][KS_HD], wf2[NT2][KS_D], wf3[NT3][KS_DI], wf4[NT4][KS_D], xf1[KS_HD];
    float gm1[KS_D][8], bt1[KS_D][8], gm2[KS_D][8], bt2[KS_D][8];
    bf16x4 res;
    f32x4 bias1, bias2, bias4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_p2 = [&]() {
        load_w<KS_D, NT2>(wf2, a.W1, ng, w, lane);
        load_affine<KS_D>(gm1, bt1, a.g1, a.be1, a.d_ln, k0d);
        bias1 = *(const f32x4*)(a.b1 + min(col, DI - 4));
    };
    auto load_p3 = [&]() {
        load_w<KS_DI, NT3>(wf3, a.W2, ng, w, lane);
        bias2 = *(const f32x4*)(a.b2 + min(col, D - 4));
    };
    auto load_p4 = [&]() {
        load_w<KS_D, NT4>(wf4, a.Wn, ng, w, lane);
        if (!HEAD) load_affine<KS_D>(gm2, bt2, a.g2, a.be2, a.d_ln, k0d);
        if (LOGITS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bias4[e] = a.bn[min(col + e, a.Nn - 1)];
        }
    };
    if (HEAD) load_p4();
    // ---------------------------------------------------------------- phase 1: z1 = vec . Wo^T + h
    if (!HEAD) {
        load_w<KS_HD, NT1>(wf1, a.Wo, ng, w, lane);
        load_x<KS_HD>(xf1, make_srd(a.vec, ((size_t)(B - 1) * a.ld_vec + HD) * 2), ((unsigned)row * a.ld_vec + k0h) * 2u, false);
        res = *(const bf16x4*)(a.h + (size_t)rowc * a.ld_h + min(col, D - 4));
        if (w >= NT1) load_p2();
        multiply<KS_HD, NT1>(wf1, xf1, red, w, lane);
        if (w < NT1 && row < B) {
            f32x4 v = tile_sum(red, w, lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bf2f(res[e]);
            store4_sc1(make_srd(a.z1, (size_t)B * D * 2), ((unsigned)row * D + col) * 2u, v);
        }
        arrive(cnt, w < NT1);
        if (w < NT1) load_p2();
        stamp();
    }
    // ---------------------------------------------------------------- phase 2: a = LN1(z1); hid = relu(a . W1^T + b1)
    if (!HEAD) {
        bf16x8 xf[KS_D];
        wait_arrivals(cnt, a.err, 1u);
        stamp();
        load_x<KS_D>(xf, make_srd(a.z1, (size_t)B * D * 2), ((unsigned)row * D + k0d) * 2u, true);
        if (w >= NT2) load_p3();
        layer_norm<KS_D>(xf, gm1, bt1, a.d_ln, a.eps1, st, w, r16, g, k0d);
        stamp();
        if (own) {
#pragma unroll
            for (int ks = 0; ks < KS_D; ++ks)
                if (ks == own_ks) *(bf16x8*)&abuf[r16][8 * (g & 1)] = xf[ks];
        }
        multiply<KS_D, NT2>(wf2, xf, red, w, lane);
        if (w < NT2 && row < B) {
            f32x4 v = tile_sum(red, w, lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bias1[e], 0.f);
            store4_sc1(make_srd(a.hid, (size_t)B * DI * 2), ((unsigned)row * DI + col) * 2u, v);
        }
        arrive(cnt + 4 * CNT_STRIDE, w < NT2);
        if (w < NT2) load_p3();
        stamp();
    }
    // ---------------------------------------------------------------- phase 3: z2 = hid . W2^T + b2 + a
    if (!HEAD) {
        bf16x8 xf[KS_DI];
        wait_arrivals(cnt + 4 * CNT_STRIDE, a.err, 2u);
        stamp();
        load_x<KS_DI>(xf, make_srd(a.hid, (size_t)B * DI * 2), ((unsigned)row * DI + k0i) * 2u, true);
        if (w >= NT3) load_p4();
        multiply<KS_DI, NT3>(wf3, xf, red, w, lane);
        if (w < NT3 && row < B) {
            f32x4 v = tile_sum(red, w, lane);
            const bf16x4 ra = *(const bf16x4*)&abuf[r16][4 * g];          // (NT3 == 1: the tile is columns 16 ng .. + 16)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bias2[e] + bf2f(ra[e]);
            store4_sc1(make_srd(a.z2, (size_t)B * D * 2), ((unsigned)row * D + col) * 2u, v);
        }
        arrive(cnt + 8 * CNT_STRIDE, w < NT3);
        if (w < NT3) load_p4();
        stamp();
    }
    // ---------------------------------------------------------------- phase 4: h' = LN2(z2); out = h' . Wn^T (+ bn)
    {
        bf16x8 xf[KS_D];
        if (HEAD) {
            // x = E[tok[row]] * sqrt(d_model) (commu/model/model.py:409-420); an id outside the vocabulary poisons its row
            // with NaN, like commu_embed_fwd
            const long long id = row < B ? a.tok[row] : 0;
            const bool bad = id < 0 || id >= a.V;
            const float* src = a.E32 + (size_t)(bad ? 0 : id) * a.d_true;
#pragma unroll
            for (int ks = 0; ks < KS_D; ++ks) {
                const int kk = k0d + 32 * ks;
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
            wait_arrivals(cnt + 8 * CNT_STRIDE, a.err, 3u);
            stamp();
            load_x<KS_D>(xf, make_srd(a.z2, (size_t)B * D * 2), ((unsigned)row * D + k0d) * 2u, true);
            layer_norm<KS_D>(xf, gm2, bt2, a.d_ln, a.eps2, st, w, r16, g, k0d);
            stamp();
        }
        if (own && row < B && a.h_out != nullptr) {
#pragma unroll
            for (int ks = 0; ks < KS_D; ++ks)
                if (ks == own_ks) st_bf16x8(a.h_out + (size_t)row * a.ld_ho + 16 * ng + 8 * (g & 1), xf[ks]);
        }
        multiply<KS_D, NT4>(wf4, xf, red, w, lane);
        if (w < NT4 && row < B && col < a.Nn && !(LOGITS && a.active != nullptr && !a.active[row])) {
            f32x4 v = tile_sum(red, w, lane);
            if (LOGITS) {
                float* o = (float*)a.out_n + (size_t)row * a.ld_on + col;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < a.Nn) o[e] = v[e] + bias4[e];
            } else {
                bf16* o = (bf16*)a.out_n + (size_t)row * a.ld_on + col;
                *(bf16x4*)o = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            }
        }
        stamp();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same four phases for WIDE models (d_model 1024, d_inner 2048, 16 heads x 64: BASELINE.json configs[4]).  A workgroup
// owns 2 .. 6 column tiles per phase and K is 1024 .. 2048, so the weights of a phase no longer fit the register file at
// once: a phase walks its tiles in BATCHES of two (two waves finish one tile each), the activation fragments stay in
// registers for the whole phase, LayerNorm parameters are read where they are used.  Hand-offs, counters, packed weight
// layout and the cache-policy discipline are those of decode_tail_kernel.  (No cross-phase weight prefetch here: the first
// batch of a phase exposes one L2 round trip.)
template <int KS>
__device__ __forceinline__ void layer_norm_wide(bf16x8 (&xf)[KS], const float* __restrict__ gamma, const float* __restrict__ beta,
                                                int d_ln, float eps, float (*st)[16], int w, int r16, int g, int k0) {
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
    for (int ks = 0; ks < KS; ++ks) {
        float gm[8], bt[8];
#pragma unroll
        for (int h4 = 0; h4 < 2; ++h4) {
            const int kk = min(k0 + 32 * ks + 4 * h4, d_ln - 4);
            const f32x4 gv = *(const f32x4*)(gamma + kk), bv = *(const f32x4*)(beta + kk);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gm[4 * h4 + e] = gv[e]; bt[4 * h4 + e] = bv[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            xf[ks][e] = f2bf(k0 + 32 * ks + e < d_ln ? (bf2f(xf[ks][e]) - mu) * rs * gm[e] + bt[e] : 0.f);
    }
}

// one phase: out tiles ng + 32 t (t < NT) of x . W^T, two tiles per batch; fin(t, v) finishes tile t (called by wave t & 1
// with the four waves' sum of the lane's 4 outputs: row r16, columns 16 (ng + 32 t) + 4 g .. + 4)
template <int KS, int NT, class Fin>
__device__ __forceinline__ void wide_phase(const bf16* __restrict__ Wp, const bf16x8 (&xf)[KS], float (*red)[8][64], int ng, int w,
                                           int lane, Fin fin) {
#pragma unroll 1
    for (int t0 = 0; t0 < NT; t0 += 2) {
        bf16x8 wf[2][KS];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tt = min(t0 + t, NT - 1);
            const bf16* wp = Wp + ((size_t)(((ng * NT + tt) * 4 + w) * KS) * 64 + lane) * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[t][ks] = ld_bf16x8(wp + ks * 512);
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = mfma16(wf[t][ks], xf[ks], acc[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[w][4 * t + e][lane] = acc[t][e];
        __syncthreads();
        if (w < 2 && t0 + w < NT) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = red[0][4 * w + e][lane] + red[1][4 * w + e][lane] + red[2][4 * w + e][lane] + red[3][4 * w + e][lane];
            fin(t0 + w, v);
        }
        __syncthreads();
    }
}

template <int D, int DI, int HD, int MODE>
__global__ __launch_bounds__(256) void decode_tail_wide_kernel(TailArgs a) {
    constexpr bool LOGITS = MODE == MODE_LOGITS, HEAD = MODE == MODE_HEAD;
    constexpr int KS_D = D / 128, KS_DI = DI / 128, KS_HD = HD / 128;
    constexpr int NT1 = D / 512, NT2 = DI / 512, NT3 = D / 512, NT4 = LOGITS ? 2 : (3 * HD + 511) / 512;
    static_assert(D % 512 == 0 && DI % 512 == 0 && HD % 128 == 0 && NT3 <= 4, "shape");
    __shared__ float red[4][8][64];
    __shared__ float st[4][16];
    __shared__ __attribute__((aligned(16))) bf16 abuf[NT3][16][16];

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
    const int ng = blockIdx.x & (NGRP - 1), mg = blockIdx.x >> 5;
    const int B = a.B, row = 16 * mg + r16;
    unsigned* cnt = a.sync + mg * CNT_STRIDE;
    if (HEAD)
        for (int i = blockIdx.x * 256 + tid; i < a.n_zero; i += gridDim.x * 256) a.zero_words[i] = 0u;
    const int k0h = w * (32 * KS_HD) + 8 * g, k0d = w * (32 * KS_D) + 8 * g, k0i = w * (32 * KS_DI) + 8 * g;
    const int rowc = min(row, B - 1);
    // the lanes that hold columns 16 (ng + 32 t) .. + 16 of a [16 x D] row block whose K range the four waves split
    auto owns = [&](int t, int& ks_out) {
        const int c0 = 16 * (ng + NGRP * t);
        ks_out = (c0 % (32 * KS_D)) / 32;
        return w == c0 / (32 * KS_D) && (g >> 1) == (c0 % 32) / 16;
    };
    // ---------------------------------------------------------------- phase 1: z1 = vec . Wo^T + h
    if (!HEAD) {
        bf16x8 xf[KS_HD];
        load_x<KS_HD>(xf, make_srd(a.vec, ((size_t)(B - 1) * a.ld_vec + HD) * 2), ((unsigned)row * a.ld_vec + k0h) * 2u, false);
        const srd_t sz1 = make_srd(a.z1, (size_t)B * D * 2);
        wide_phase<KS_HD, NT1>(a.Wo, xf, red, ng, w, lane, [&](int t, f32x4 v) {
            const int col = 16 * (ng + NGRP * t) + 4 * g;
            if (row < B) {
                const bf16x4 res = *(const bf16x4*)(a.h + (size_t)rowc * a.ld_h + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bf2f(res[e]);
                store4_sc1(sz1, ((unsigned)row * D + col) * 2u, v);
            }
        });
        arrive(cnt, w < 2);
    }
    // ---------------------------------------------------------------- phase 2: a = LN1(z1); hid = relu(a . W1^T + b1)
    if (!HEAD) {
        bf16x8 xf[KS_D];
        wait_arrivals(cnt, a.err, 1u);
        load_x<KS_D>(xf, make_srd(a.z1, (size_t)B * D * 2), ((unsigned)row * D + k0d) * 2u, true);
        layer_norm_wide<KS_D>(xf, a.g1, a.be1, a.d_ln, a.eps1, st, w, r16, g, k0d);
#pragma unroll
        for (int t = 0; t < NT3; ++t) {
            int oks;
            if (owns(t, oks)) {
#pragma unroll
                for (int ks = 0; ks < KS_D; ++ks)
                    if (ks == oks) *(bf16x8*)&abuf[t][r16][8 * (g & 1)] = xf[ks];
            }
        }
        const srd_t shid = make_srd(a.hid, (size_t)B * DI * 2);
        wide_phase<KS_D, NT2>(a.W1, xf, red, ng, w, lane, [&](int t, f32x4 v) {
            const int col = 16 * (ng + NGRP * t) + 4 * g;
            if (row < B) {
                const f32x4 b = *(const f32x4*)(a.b1 + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + b[e], 0.f);
                store4_sc1(shid, ((unsigned)row * DI + col) * 2u, v);
            }
        });
        arrive(cnt + 4 * CNT_STRIDE, w < 2);
    }
    // ---------------------------------------------------------------- phase 3: z2 = hid . W2^T + b2 + a
    if (!HEAD) {
        bf16x8 xf[KS_DI];
        wait_arrivals(cnt + 4 * CNT_STRIDE, a.err, 2u);
        load_x<KS_DI>(xf, make_srd(a.hid, (size_t)B * DI * 2), ((unsigned)row * DI + k0i) * 2u, true);
        const srd_t sz2 = make_srd(a.z2, (size_t)B * D * 2);
        wide_phase<KS_DI, NT3>(a.W2, xf, red, ng, w, lane, [&](int t, f32x4 v) {
            const int col = 16 * (ng + NGRP * t) + 4 * g;
            if (row < B) {
                const f32x4 b = *(const f32x4*)(a.b2 + col);
                const bf16x4 ra = *(const bf16x4*)&abuf[t][r16][4 * g];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += b[e] + bf2f(ra[e]);
                store4_sc1(sz2, ((unsigned)row * D + col) * 2u, v);
            }
        });
        arrive(cnt + 8 * CNT_STRIDE, w < 2);
    }
    // ---------------------------------------------------------------- phase 4: h' = LN2(z2); out = h' . Wn^T (+ bn)
    {
        bf16x8 xf[KS_D];
        if (HEAD) {
            const long long id = row < B ? a.tok[row] : 0;
            const bool bad = id < 0 || id >= a.V;
            const float* src = a.E32 + (size_t)(bad ? 0 : id) * a.d_true;
#pragma unroll
            for (int ks = 0; ks < KS_D; ++ks) {
                const int kk = k0d + 32 * ks;
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
            wait_arrivals(cnt + 8 * CNT_STRIDE, a.err, 3u);
            load_x<KS_D>(xf, make_srd(a.z2, (size_t)B * D * 2), ((unsigned)row * D + k0d) * 2u, true);
            layer_norm_wide<KS_D>(xf, a.g2, a.be2, a.d_ln, a.eps2, st, w, r16, g, k0d);
        }
        if (row < B && a.h_out != nullptr) {
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
                int oks;
                if (owns(t, oks)) {
#pragma unroll
                    for (int ks = 0; ks < KS_D; ++ks)
                        if (ks == oks) st_bf16x8(a.h_out + (size_t)row * a.ld_ho + 16 * (ng + NGRP * t) + 8 * (g & 1), xf[ks]);
                }
            }
        }
        wide_phase<KS_D, NT4>(a.Wn, xf, red, ng, w, lane, [&](int t, f32x4 v) {
            const int col = 16 * (ng + NGRP * t) + 4 * g;
            if (row < B && col < a.Nn && !(LOGITS && a.active != nullptr && !a.active[row])) {
                if (LOGITS) {
                    float* o = (float*)a.out_n + (size_t)row * a.ld_on + col;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < a.Nn) o[e] = v[e] + a.bn[min(col + e, a.Nn - 1)];
                } else {
                    bf16* o = (bf16*)a.out_n + (size_t)row * a.ld_on + col;
                    *(bf16x4*)o = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                }
            }
        });
    }
}

}  // namespace

extern "C" int commu_decode_tail_supported(int B, int D, int DI, int HD) {
    if (B < 1 || B > 64) return 0;
    if (D == 512 && DI == 1024 && (HD == 512 || HD == 640)) return 1;
    if (D == 1024 && DI == 2048 && HD == 1024) return 1;          // decode_tail_wide_kernel
    return 0;
}

extern "C" int commu_decode_tail_sync_words(void) { return 12 * CNT_STRIDE; }

static unsigned long long* g_tail_trace = nullptr;
static TailArgs tail_args_zero() {
    TailArgs a;
    memset(&a, 0, sizeof(a));
    a.trace = g_tail_trace;
    return a;
}
/* diagnostics: every following layer-tail / head launch writes the 100 MHz timestamps of its phase boundaries to
 * buf[workgroup][16] (null: off) */
extern "C" int commu_decode_tail_trace(unsigned long long* buf) {
    g_tail_trace = buf;
    return 0;
}

/* bytes of the packed copy of a [N][K] weight (K % 128 == 0) */
extern "C" long long commu_decode_tail_pack_bytes(int N, int K) {
    if (N < 1 || K < 128 || (K % 128)) return 0;
    const int nt = ((N + 15) / 16 + NGRP - 1) / NGRP;
    return (long long)NGRP * nt * 4 * (K / 128) * 1024;
}

extern "C" int commu_decode_tail_pack(const void* W, int ldw, int N, int K, void* out, hipStream_t stream) {
    const long long bytes = commu_decode_tail_pack_bytes(N, K);
    if (bytes == 0 || (ldw % 8)) return -22;
    const long long nchunks = bytes / 16;
    const int nt = ((N + 15) / 16 + NGRP - 1) / NGRP;
    COMMU_LAUNCH(pack_weight_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, stream, (const bf16*)W, ldw, N,
                 K / 128, nt, (bf16*)out, nchunks);
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_decode_layer_tail(const void* vec, int ld_vec, const void* h, int ld_h, const void* Wo_packed,
                                       const void* W1_packed, const float* b1, const void* W2_packed, const float* b2,
                                       const float* g1, const float* be1, float eps1, const float* g2, const float* be2,
                                       float eps2, int d_ln, const void* Wn_packed, int Nn, const float* bn, int logits,
                                       const unsigned char* active, void* z1, void* hid, void* z2, void* h_out,
                                       int ld_ho, void* out_n, int ld_on, int B, int D, int DI, int HD, unsigned* sync,
                                       unsigned* err, hipStream_t stream) {
    if (!commu_decode_tail_supported(B, D, DI, HD)) return -22;
    if (d_ln < 4 || d_ln > D || (d_ln % 4) || (ld_vec % 8) || (ld_h % 4) || (ld_ho % 8) || (ld_on % 4) || sync == nullptr ||
        err == nullptr)
        return -22;
    if (logits ? (Nn < 1 || Nn > 1024 || bn == nullptr) : (Nn != 3 * HD)) return -22;
    TailArgs a = tail_args_zero();
    a.vec = (const bf16*)vec; a.ld_vec = ld_vec;
    a.h = (const bf16*)h; a.ld_h = ld_h;
    a.Wo = (const bf16*)Wo_packed; a.W1 = (const bf16*)W1_packed; a.W2 = (const bf16*)W2_packed;
    a.Wn = (const bf16*)Wn_packed; a.Nn = Nn;
    a.b1 = b1; a.b2 = b2; a.bn = bn;
    a.g1 = g1; a.be1 = be1; a.g2 = g2; a.be2 = be2;
    a.eps1 = eps1; a.eps2 = eps2; a.d_ln = d_ln;
    a.z1 = (bf16*)z1; a.hid = (bf16*)hid; a.z2 = (bf16*)z2;
    a.h_out = (bf16*)h_out; a.ld_ho = ld_ho;
    a.out_n = out_n; a.ld_on = ld_on;
    a.B = B; a.sync = sync; a.err = err; a.active = active;
    const dim3 grid(NGRP * ((B + 15) / 16));
    if (D == 1024) {
        if (logits) COMMU_LAUNCH((decode_tail_wide_kernel<1024, 2048, 1024, MODE_LOGITS>), grid, dim3(256), 0, stream, a);
        else COMMU_LAUNCH((decode_tail_wide_kernel<1024, 2048, 1024, MODE_QKV>), grid, dim3(256), 0, stream, a);
    } else if (HD == 512) {
        if (logits) COMMU_LAUNCH((decode_tail_kernel<512, 1024, 512, MODE_LOGITS>), grid, dim3(256), 0, stream, a);
        else COMMU_LAUNCH((decode_tail_kernel<512, 1024, 512, MODE_QKV>), grid, dim3(256), 0, stream, a);
    } else {          // 10 heads x 64: the released default config (d_model 500, 10 x 50) after zero padding
        if (logits) COMMU_LAUNCH((decode_tail_kernel<512, 1024, 640, MODE_LOGITS>), grid, dim3(256), 0, stream, a);
        else COMMU_LAUNCH((decode_tail_kernel<512, 1024, 640, MODE_QKV>), grid, dim3(256), 0, stream, a);
    }
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_decode_head(const int64_t* tok, const float* E, int d_true, int V, float scale,
                                 const void* Wqkv_packed, void* h_out, int ld_ho, void* qkv, int ld_qkv, int B, int D,
                                 int DI, int HD, unsigned* zero_words, int n_zero, hipStream_t stream) {
    if (!commu_decode_tail_supported(B, D, DI, HD)) return -22;
    if (d_true <= 0 || d_true > D || (d_true % 4) || (ld_ho % 8) || (ld_qkv % 4) || h_out == nullptr) return -22;
    TailArgs a = tail_args_zero();
    a.tok = tok; a.E32 = E; a.d_true = d_true; a.V = V; a.emb_scale = scale;
    a.Wn = (const bf16*)Wqkv_packed; a.Nn = 3 * HD;
    a.h_out = (bf16*)h_out; a.ld_ho = ld_ho;
    a.out_n = qkv; a.ld_on = ld_qkv;
    a.B = B; a.zero_words = zero_words; a.n_zero = zero_words != nullptr ? n_zero : 0;
    const dim3 grid(NGRP * ((B + 15) / 16));
    if (D == 1024) COMMU_LAUNCH((decode_tail_wide_kernel<1024, 2048, 1024, MODE_HEAD>), grid, dim3(256), 0, stream, a);
    else if (HD == 512) COMMU_LAUNCH((decode_tail_kernel<512, 1024, 512, MODE_HEAD>), grid, dim3(256), 0, stream, a);
    else COMMU_LAUNCH((decode_tail_kernel<512, 1024, 640, MODE_HEAD>), grid, dim3(256), 0, stream, a);
    COMMU_LAUNCH_CHECK();
    return 0;
}
