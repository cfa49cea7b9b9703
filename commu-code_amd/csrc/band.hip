// Fused band pass of the relative-position attention backward (gfx950): ONE sweep over dS-by-distance produces both
//   dq[:, h]  = dq_ac[:, h] + dSk[h] . Rd[:, h]            (BD part of the query gradient)
//   dRd[:, h] += dSk[h]^T . qv[:, h]                       (gradient of r_net's output, summed over tokens)
// (autograd of BD = rel_shift((q + r_r_bias) . r^T), commu/model/model.py:316-318,251-259; dSk[m][d] = dS of token row
// m = i*B + b at distance d = i + M - j, stored by relattn_bwd_q as [H][T*B/64][ld_dsk/128] TILES of [64 rows][128
// distances]: the 16 KB this kernel streams per step are contiguous in HBM).  It replaces a per-head NT GEMM plus a per-head TN GEMM that each streamed
// the 0.5-GB dSk from HBM; here every dSk byte is read once and feeds 128 FLOP.
//
// Workgroup = (head, pair of token slices s and 2P-1-s: the causal band makes late slices long and early ones short,
// the pair is balanced).  512 threads.  Per 64-token step and 256-distance chunk (32 KB of dSk + the matching
// 32 KB of Rd[:, h], staged by LDS-DMA, double buffered):
//   * dRd: wave w owns distances 32 w .. 32 w + 31 of the chunk, all 64 features: 16 MFMA, accumulators for all four
//     chunks stay in registers for the whole kernel (a [1024 x 64] fp32 slab per workgroup, reduced afterwards);
//   * dq:  wave w owns rows 16 (w >> 1) .. + 15, features 32 (w & 1) .. + 31: 16 MFMA over the chunk's 256 distances.
// Operand fragments whose contraction index is the LDS image's row index come from ds_read_b64_tr_b16; 32-byte units
// are XOR-swizzled so both the transpose reads and the ds_read_b128 of the dSk image are conflict-free.
// The kernel is HBM-bound (the chip streams dSk at ~5 TB/s = 32 KB per CU every ~3700 cycles; the MFMA work of a
// chunk is ~1000 cycles), so the pipeline is a plain two-stage one: stage next, compute current, one barrier per chunk.
#include "common.h"
#include "commu_hip.h"
#include <stdlib.h>

namespace {

typedef __amdgpu_buffer_rsrc_t srd_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int CH = 128;                  // distances per streamed chunk
constexpr int SBYTES = 64 * CH * 2;      // dSk chunk image [64 m][128 d] (16 KB)
constexpr int RHALF = 512;               // distances of Rd[:, h] resident per pass
constexpr int RBYTES = RHALF * 64 * 2;   // Rd half image [512 d][64 f] (64 KB)
constexpr int QBYTES = 64 * 64 * 2;      // qv step image [64 m][64 f] (8 KB)
constexpr int KMAX = 4096;               // largest K: passes of RHALF = 512 distances (4 chunks each), any number of them
constexpr int NRING = 3;                 // dSk chunk buffers: one being read, two in flight
constexpr int NSTEPBUF = 3;              // per-step tiles (qv, residual dq rows): steps with a chunk in the ring

struct BandArgs {
    const bf16* dsk; int ld_dsk; long long dsk_hstride;
    const bf16* rd; int ld_rd;
    const bf16* qv; int ld_qv;
    const bf16* dq_ac; int ld_ac;
    bf16* dq; int ld_dq;
    float* slabs;               // [H][P][KMAX][64]
    int TB, K, H, tri_B, tri_M, nsteps, P, sps;          // sps = 64-row steps per slice
    int abl;                    // profiling ablations (COMMU_BAND_ABL): 1 no residual load, 2 no staging in the loop, 4 no MFMA
};

__device__ __forceinline__ void dma16(srd_t srd, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

struct It {          // position in the workgroup's flattened (sub-slice, step, chunk of this pass) sequence
    int sub, st, c, nch, valid, qb;          // qb: which of the per-step tile buffers this step uses
};

constexpr int RD_OFF = 0;                                   // Rd half, resident for a whole pass
constexpr int S_OFF = RBYTES;                               // ring of dSk chunk buffers
constexpr int Q_OFF = S_OFF + NRING * SBYTES;               // qv step tiles
constexpr int A_OFF = Q_OFF + NSTEPBUF * QBYTES;            // residual step tiles: dq_ac (pass 0) / pass-0 dq (pass 1), [64 m][64 f]

__global__ __launch_bounds__(512) void band_bwd_kernel(const BandArgs a) {
    __shared__ __attribute__((aligned(1024))) char smem[RBYTES + NRING * SBYTES + 2 * NSTEPBUF * QBYTES];          // 160 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int h = (int)blockIdx.x % a.H, pair = (int)blockIdx.x / a.H;

    const bf16* dskh = a.dsk + (long long)h * a.dsk_hstride;
    const srd_t sS = __builtin_amdgcn_make_buffer_rsrc((void*)dskh, 0, (int)(unsigned)((size_t)a.TB * a.ld_dsk * 2), 0x00020000);
    const srd_t sR = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rd + h * 64), 0,
                                                       (int)(unsigned)(((size_t)(a.K - 1) * a.ld_rd + 64) * 2), 0x00020000);
    const srd_t sQ = __builtin_amdgcn_make_buffer_rsrc((void*)(a.qv + h * 64), 0,
                                                       (int)(unsigned)(((size_t)(a.TB - 1) * a.ld_qv + 64) * 2), 0x00020000);
    const srd_t sA0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dq_ac + h * 64), 0,
                                                        (int)(unsigned)(((size_t)(a.TB - 1) * a.ld_ac + 64) * 2), 0x00020000);
    const srd_t sA1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dq + h * 64), 0,
                                                        (int)(unsigned)(((size_t)(a.TB - 1) * a.ld_dq + 64) * 2), 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
    const LDS_AS char* lds = (const LDS_AS char*)smem;

    // ---- fragment addressing
    auto tr8 = [&](int off, int hstride) -> bf16x8 {          // 8 consecutive image rows of one image column
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(lds + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(lds + off + hstride));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    const int fi = ((r16 >> 2) & 3) | ((g & 1) << 2);           // unit swizzle of the rows a transpose read touches (dSk image)
    const int g2 = ((r16 >> 3) & 1) | ((g & 1) << 1);           // same for the 128-byte-pitch images
    const int baseS = (8 * g + (r16 >> 2)) * 256 + (r16 & 3) * 8;
    const int baseR = (8 * g + (r16 >> 2)) * 128 + (r16 & 3) * 8;
    const int sTN = baseS + ((w ^ fi) << 5);                    // dSk^T fragment (TN, B operand): distance block w of the chunk
    int qTN[4], rNT[2], sNT[4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) qTN[fb] = baseR + ((fb ^ g2) << 5);                      // qv^T fragments (TN, A operand)
#pragma unroll
    for (int j = 0; j < 2; ++j) rNT[j] = baseR + (((2 * (w & 1) + j) ^ g2) << 5);           // Rd^T fragments (NT, A operand)
    {
        const int fl = (r16 & 3) | (((r16 >> 3) & 1) << 2), X = fl << 1;
        const int rowb = ((w >> 1) * 16 + r16) * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) sNT[j] = rowb + ((((j * 4) ^ (X & 12)) | (g ^ (X & 3))) << 4);          // dSk rows (NT, B operand)
    }

    f32x4 accR[4][4];          // dRd[chunk of the pass][feature block] of distance block w: D[f = 4g + reg][d = r16]
    const int kpad = ((a.K + RHALF - 1) / RHALF) * RHALF;          // slab rows
    float* slab = a.slabs + ((size_t)h * a.P + pair) * (size_t)kpad * 64;

    // Two passes over the workgroup's token rows: pass p keeps distances 512 p .. 512 p + 511 of Rd[:, h] RESIDENT in LDS
    // (64 KB) and streams only dSk, 16 KB per chunk through a ring of three buffers (two chunks in flight).  With a
    // step's first chunk come its qv tile and the residual rows the step's dq is added to (dq_ac in pass 0; pass 1 adds
    // to what pass 0 stored): everything enters through LDS-DMA, so no register load ever forces a full vmcnt drain.
#define BAND_TN(C)                                                                                                   \
    {                                                                                                                \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                           \
            const bf16x8 bs = tr8(so + sTN + ks * 8192, 1024);                                                       \
            _Pragma("unroll") for (int fb = 0; fb < 4; ++fb) accR[C][fb] = mfma16(aq[fb][ks], bs, accR[C][fb]);      \
        }                                                                                                            \
    }
#define BAND_PASS(PASS)                                                                                              \
    {                                                                                                                \
        _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) accR[c][j] = (f32x4){0.f, 0.f, 0.f, 0.f};                  \
        auto slice_of = [&](int sub) { return sub == 0 ? pair : 2 * a.P - 1 - pair; };                               \
        auto nchunks = [&](int st) {          /* chunks of THIS pass that step st reaches (0 .. 4) */                \
            int keff = a.K;                                                                                          \
            if (a.tri_B > 0) keff = min(a.K, (st * 64 + 63) / a.tri_B + a.tri_M + 1);                                \
            return max(0, min(4, (keff + CH - 1) / CH - 4 * (PASS)));                                                \
        };                                                                                                           \
        auto seek = [&](It& it) {          /* first (sub, st) at or after the current one with work in this pass */  \
            for (; it.sub < 2; ++it.sub, it.st = -1) {                                                               \
                const int s0 = slice_of(it.sub) * a.sps, send = min(a.nsteps, s0 + a.sps);                           \
                if (it.st < s0) it.st = s0;                                                                          \
                for (; it.st < send; ++it.st) {                                                                      \
                    it.nch = nchunks(it.st);                                                                         \
                    if (it.nch > 0) { it.c = 0; it.valid = 1; it.qb = it.qb == NSTEPBUF - 1 ? 0 : it.qb + 1; return; }                     \
                }                                                                                                    \
            }                                                                                                        \
            it.valid = 0;                                                                                            \
        };                                                                                                           \
        auto advance = [&](It& it) {                                                                                 \
            if (!it.valid || ++it.c < it.nch) return;                                                                \
            ++it.st;                                                                                                 \
            seek(it);                                                                                                \
        };                                                                                                           \
        auto stage = [&](const It& it, int slot_) {                                                                  \
            int ln = lane;                                                                                           \
            asm volatile("" : "+v"(ln));          /* (per-lane offsets recomputed at every use, not kept in VGPRs) */\
            const unsigned dS = lds0 + S_OFF + slot_ * SBYTES + w * 2048;                                            \
            /* dsk is tiled: chunk (step, c) = 16 contiguous KB, [64 rows][128 distances] */                          \
            const unsigned soS = ((unsigned)it.st * (unsigned)(a.ld_dsk >> 7) + (unsigned)(4 * (PASS) + it.c)) * (unsigned)SBYTES; \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {          /* piece = 4 rows x 256 bytes = 1 contiguous KB */ \
                const int m = 4 * (2 * w + j) + (ln >> 4), sl = ln & 15;                                             \
                const int f = (m & 3) | (((m >> 3) & 1) << 2);                                                       \
                dma16(sS, (unsigned)(m * 256 + (sl ^ (f << 1)) * 16), soS, dS + j * 1024);                           \
            }                                                                                                        \
            if (it.c == 0) {                                                                                         \
                const int r = 8 * w + (ln >> 3), sl = ln & 7;                                                        \
                const int q2 = ((r >> 1) & 1) | (((r >> 3) & 1) << 1);                                               \
                dma16(sQ, (unsigned)(r * a.ld_qv + (sl ^ (q2 << 1)) * 8) * 2u,                                       \
                      (unsigned)(it.st * 64) * (unsigned)a.ld_qv * 2u, lds0 + Q_OFF + it.qb * QBYTES + w * 1024);   \
                if ((PASS) == 0)                                                                                     \
                    dma16(sA0, (unsigned)(r * a.ld_ac + sl * 8) * 2u, (unsigned)(it.st * 64) * (unsigned)a.ld_ac * 2u, \
                          lds0 + A_OFF + it.qb * QBYTES + w * 1024);                                                 \
                else                                                                                                 \
                    dma16(sA1, (unsigned)(r * a.ld_dq + sl * 8) * 2u, (unsigned)(it.st * 64) * (unsigned)a.ld_dq * 2u, \
                          lds0 + A_OFF + it.qb * QBYTES + w * 1024);                                                 \
            }                                                                                                        \
        };                                                                                                           \
        It q0, q1, q2_;                                                                                              \
        q0.sub = 0; q0.st = -1; q0.qb = NSTEPBUF - 1; q0.c = 0; q0.nch = 0; q0.valid = 0;                            \
        seek(q0);                                                                                                    \
        q1 = q0; advance(q1);                                                                                        \
        q2_ = q1; advance(q2_);                                                                                      \
        if (q0.valid) {                                                                                              \
            /* this pass's half of Rd[:, h]: 64 pieces of 8 rows x 128 bytes, 8 per wave */                          \
            int ln = lane;                                                                                           \
            asm volatile("" : "+v"(ln));                                                                             \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                          \
                const int r = 8 * (8 * w + j) + (ln >> 3), sl = ln & 7;                                              \
                const int q2 = ((r >> 1) & 1) | (((r >> 3) & 1) << 1);                                               \
                dma16(sR, (unsigned)(r * a.ld_rd + (sl ^ (q2 << 1)) * 8) * 2u,                                       \
                      (unsigned)(RHALF * (PASS)) * (unsigned)a.ld_rd * 2u, lds0 + RD_OFF + (8 * w + j) * 1024);      \
            }                                                                                                        \
            stage(q0, 0);                                                                                            \
            if (q1.valid) stage(q1, 1);                                                                              \
        }                                                                                                            \
        if (q1.valid) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                               \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        __syncthreads();                                                                                             \
        f32x4 accQ[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};                                  \
        bf16x8 aq[4][2];          /* qv^T fragments of the current step (read once per step) */                     \
        int slot = 0;                                                                                                \
        while (q0.valid) {                                                                                           \
            int extra = 0;                                                                                           \
            if (q2_.valid && !(a.abl & 2)) stage(q2_, slot == 0 ? 2 : slot - 1);          /* the buffer read in the previous iteration */ \
            const int so = S_OFF + slot * SBYTES, ro = RD_OFF + q0.c * (CH * 128), qo = Q_OFF + q0.qb * QBYTES;      \
            if (q0.c == 0) {                                                                                         \
                _Pragma("unroll") for (int fb = 0; fb < 4; ++fb)                                                     \
                    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) aq[fb][ks] = tr8(qo + qTN[fb] + ks * 4096, 512);\
            }                                                                                                        \
            /* dRd += dSk^T . qv for this chunk (static accumulator index: one copy of the body per chunk) */        \
            if (!(a.abl & 4)) switch (q0.c) {                                                                        \
                case 0: BAND_TN(0) break;                                                                            \
                case 1: BAND_TN(1) break;                                                                            \
                case 2: BAND_TN(2) break;                                                                            \
                default: BAND_TN(3) break;                                                                           \
            }                                                                                                        \
            /* dq += dSk . Rd over this chunk's 128 distances */                                                     \
            if (!(a.abl & 4)) _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                     \
                const bf16x8 bs = *(const LDS_AS bf16x8*)(lds + so + sNT[kk]);                                       \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                      \
                    const bf16x8 ar = tr8(ro + rNT[j] + kk * 4096, 512);                                             \
                    accQ[j] = mfma16(ar, bs, accQ[j]);                                                               \
                }                                                                                                    \
            }                                                                                                        \
            if (q0.c == q0.nch - 1) {          /* last chunk of the step in this pass: dq rows = residual + this pass */ \
                const int lrow = (w >> 1) * 16 + r16;                                                                \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                      \
                    const int fl_ = (2 * (w & 1) + j) * 16 + 4 * g;                                                  \
                    bf16x4 r4 = (bf16x4){0, 0, 0, 0};                                                                \
                    if (!(a.abl & 1)) r4 = *(const LDS_AS bf16x4*)(lds + A_OFF + q0.qb * QBYTES + lrow * 128 + fl_ * 2); \
                    bf16x4 o;                                                                                        \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] = f2bf(accQ[j][e] + bf2f(r4[e]));             \
                    *(bf16x4*)(a.dq + (size_t)(q0.st * 64 + lrow) * a.ld_dq + h * 64 + fl_) = o;                     \
                    accQ[j] = (f32x4){0.f, 0.f, 0.f, 0.f};                                                           \
                }                                                                                                    \
                extra = 2;                                                                                           \
            }                                                                                                        \
            /* the next chunk (and its step tiles) must have landed; the chunk staged in this iteration and this      \
               iteration's dq stores stay in flight (a chunk group has at least two pieces) */                        \
            {                                                                                                        \
                const int keep = (q2_.valid ? 2 : 0) + extra;                                                        \
                switch (keep) {                                                                                      \
                    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;                                  \
                    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;                                  \
                    default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;                                 \
                }                                                                                                    \
            }                                                                                                        \
            q0 = q1; q1 = q2_;                                                                                       \
            advance(q2_);                                                                                            \
            slot = slot == 2 ? 0 : slot + 1;                                                                                   \
            __syncthreads();                                                                                         \
        }                                                                                                            \
        /* dRd slab rows of this pass: [512][64] fp32, lane holds slab[d = .. + r16][f = .. + 4g + reg] */           \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                              \
            const int d = RHALF * (PASS) + c * CH + w * 16 + r16;                                                    \
            _Pragma("unroll") for (int fb = 0; fb < 4; ++fb)                                                         \
                *(f32x4*)(slab + (size_t)d * 64 + fb * 16 + 4 * g) = accR[c][fb];                                    \
        }                                                                                                            \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          /* (the next pass counts its own loads from zero) */ \
        __syncthreads();                                                                                             \
    }
    const int npass = kpad / RHALF;
    for (int pass = 0; pass < npass; ++pass) BAND_PASS(pass)
#undef BAND_TN
#undef BAND_PASS
}

}  // namespace

extern "C" int commu_attn_band_slabs(int T, int B) {
    // token-slice pairs per head (the slab count of commu_relattn_bwd_band); 0: shape not taken by the fused kernel
    const long long TB = (long long)T * B;
    if (TB % 64 != 0 || TB < 8192) return 0;
    return 32;
}

extern "C" int commu_attn_band_pairs(int T, int B, int H) {
    // H * P workgroups (one per CU: 160 KB of LDS) fill the chip once; every workgroup flushes a [K x 64] fp32 slab, so more
    // pairs than that only add slab bytes (16 heads: 32 pairs wrote 0.54 GB per layer at K = 4096, 16 pairs write half)
    if (commu_attn_band_slabs(T, B) == 0 || H <= 0) return 0;
    const int P = 256 / H;
    return P < 1 ? 1 : (P > 32 ? 32 : P);
}

extern "C" int commu_relattn_bwd_band(const void* dsk, int ld_dsk, const void* rd, int ld_rd, const void* qv2, int ld_qv,
                                      const void* dq_ac, int ld_ac, void* dq, int ld_dq, float* slabs, int T, int M,
                                      int B, int H, int DH, int band, hipStream_t stream) {
    const long long TB = (long long)T * B;
    const int K = T + M, P = commu_attn_band_pairs(T, B, H);
    if (P == 0 || DH != 64 || K > KMAX || ld_dsk < K || (ld_dsk % 128) || (ld_rd % 8) || (ld_qv % 8) || (ld_ac % 4) ||
        (ld_dq % 4) || (size_t)TB * ld_dsk * 2 >= 0x7FFF0000ull)
        return -22;
    BandArgs a;
    a.dsk = (const bf16*)dsk; a.ld_dsk = ld_dsk; a.dsk_hstride = TB * ld_dsk;
    a.rd = (const bf16*)rd; a.ld_rd = ld_rd;
    a.qv = (const bf16*)qv2; a.ld_qv = ld_qv;
    a.dq_ac = (const bf16*)dq_ac; a.ld_ac = ld_ac;
    a.dq = (bf16*)dq; a.ld_dq = ld_dq;
    a.slabs = slabs;
    a.TB = (int)TB; a.K = K; a.H = H; a.tri_B = band ? B : 0; a.tri_M = M;
    a.nsteps = (int)(TB / 64); a.P = P; a.sps = (a.nsteps + 2 * P - 1) / (2 * P);
    a.abl = 0;          // (profiling ablations are compiled in but never enabled from the product path)
    COMMU_LAUNCH(band_bwd_kernel, dim3(H * P), dim3(512), 0, stream, a);
    COMMU_LAUNCH_CHECK();
    return 0;
}
