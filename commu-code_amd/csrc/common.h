// Shared device helpers for the ComMU Transformer-XL hot-path kernels (gfx950 / CDNA4 only).
// wave = 64 lanes; MFMA fragments follow the 16x16x32 bf16 layout:
//   A operand: lane l holds A[row = l&15][k = 8*(l>>4) .. +8]
//   B operand: lane l holds B[k = 8*(l>>4) .. +8][col = l&15]
//   C/D      : lane l holds C[row = 4*(l>>4) + reg][col = l&15], reg = 0..3
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

#define WAVE 64
#define LDS_AS __attribute__((address_space(3)))

// clear any stale (sticky-less) runtime error left by other users of the HIP runtime in this
// thread, then launch; COMMU_LAUNCH_CHECK() afterwards reports only our own launch failures
#define COMMU_LAUNCH(...)                 \
    do {                                  \
        (void)hipGetLastError();          \
        hipLaunchKernelGGL(__VA_ARGS__);  \
    } while (0)

#define COMMU_LAUNCH_CHECK()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// reduce across the 16 lanes that share (lane >> 4) with DPP (VALU speed; __shfl_xor would go
// through ds_bpermute).  quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140: after the four steps every lane of the row holds the full reduction.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}

// XCD-aware block remap (8 XCDs, block b runs on XCD b % 8): give each XCD a contiguous
// range of logical tile ids so neighbouring tiles share an L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// K16 dropout: counter-based mask, identical in forward and backward.  keep(seed, idx) is a 32-bit
// integer hash (lowbias32, keyed by the seed) of the element index compared with p * 2^32; kept values are scaled by
// 1/(1-p) (nn.Dropout semantics, commu/model/model.py:166,168,210-211,454,585-586,601).
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// keyed form: the seed enters BETWEEN the two multiply rounds, so the masks of two seeds are related by a
// pseudo-random permutation of the index space, never by a shift of it (with mix32(idx + seed) two dropout sites whose
// seeds differ by less than the tensor size would have used shifted copies of one mask)
__device__ __forceinline__ unsigned mix32k(unsigned idx, unsigned key) {
    unsigned x = idx;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= key; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// (the key itself is mixed first -- wave-uniform, so it costs scalar instructions once -- because seeds that differ
//  only in a few low bits would otherwise change the second round's input by a near-constant)
// Element-wise dropout sites (round 4): ONE hash word per TWO consecutive element indices, on full-rate 24-bit multiply-adds
// (the keyed lowbias32 above costs two quarter-rate 32-bit multiplies per ELEMENT: a GEMM epilogue at K = 512 could not hide
// it, +17-25 us per launch).  word(q = idx >> 1) = two multiply rounds around a xor-shift, keyed by mix32(seed) (scalar work);
// the even index takes the low 16 bits, the odd one the high 16, compared with thr16 = round(p * 65536); kept values are
// scaled by 1 / (1 - thr16 / 65536), the exact keep probability.  Host mirror: ops.dropout_keep_mask.
struct DropKey {
    unsigned key, k2;
};
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu) + c; }
__device__ __forceinline__ DropKey drop_key(unsigned seed) {
    DropKey k;
    k.key = mix32(seed);
    k.k2 = k.key * 0x85EBCA6Bu + 0x6A09E667u;
    return k;
}
__device__ __forceinline__ unsigned drop_word(unsigned q, DropKey k) {
    unsigned y = mad24(q >> 24, 0xB5297Au, mad24(q, 0xD2B74Bu, k.key));
    y ^= y >> 13;
    const unsigned w = mad24(y, 0x9E3779u, k.k2);
    return w ^ (w >> 15);
}
__device__ __forceinline__ bool drop_half(unsigned w, unsigned odd, unsigned thr16) {
    return (odd ? (w >> 16) : (w & 0xFFFFu)) >= thr16;
}
__device__ __forceinline__ bool drop_keep(DropKey k, unsigned idx, unsigned thr16) {
    return drop_half(drop_word(idx >> 1, k), idx & 1u, thr16);
}
__device__ __forceinline__ bool drop_keep(unsigned seed, unsigned idx, unsigned thr16) { return drop_keep(drop_key(seed), idx, thr16); }
// host side: 16-bit threshold of a rate (0: off) and the exact keep scale
static inline unsigned drop_threshold16(float p) {
    if (p <= 0.f) return 0u;
    const unsigned t = (unsigned)((double)p * 65536.0 + 0.5);
    return t < 1u ? 1u : (t > 65535u ? 65535u : t);
}
static inline float drop_keep_scale16(unsigned thr16) { return 1.f / (1.f - (float)thr16 / 65536.f); }

// Seed salt (one copy per translation unit): every dropout site uses `seed argument + g_seed_salt`.  Eager launches never
// touch it (0: the seed argument alone decides).  A training step replayed from a hipGraph has its seed ARGUMENTS
// frozen at capture time, so the step's first nodes (commu_set_seed_salt: one tiny kernel per translation unit) load a
// fresh salt from device memory and every replay draws new masks, forward and backward of a replay agreeing.
static __device__ unsigned g_seed_salt = 0u;
__device__ __forceinline__ unsigned salted(unsigned seed) { return seed + g_seed_salt; }
static __global__ void set_seed_salt_kernel(const unsigned* __restrict__ src) { g_seed_salt = src ? *src : 0u; }
#define COMMU_DEFINE_SEED_SALT_SETTER(NAME)                                                     \
    int NAME(const unsigned* src, hipStream_t stream) {                                         \
        COMMU_LAUNCH(set_seed_salt_kernel, dim3(1), dim3(1), 0, stream, src);                   \
        COMMU_LAUNCH_CHECK();                                                                   \
        return 0;                                                                               \
    }

__device__ __forceinline__ bf16x8 ld_bf16x8(const bf16* p) { return *(const bf16x8*)p; }
__device__ __forceinline__ void st_bf16x8(bf16* p, bf16x8 v) { *(bf16x8*)p = v; }
