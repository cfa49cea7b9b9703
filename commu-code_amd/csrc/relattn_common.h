// Shared declarations of the relative-position attention kernel families (relattn.hip: 16x16 MFMA layout;
// relattn3.hip: 32x32 MFMA / transposed-score layout).  Device helpers live in an anonymous namespace: every
// translation unit gets its own copy.
#pragma once
#include "common.h"
#include "commu_hip.h"

struct AttnArgs {
    const bf16* q;      // rows i in [0,T)   : q  + (i*B+b)*ld_qkv + h*DH        (forward only)
    const bf16* k;      // rows j in [0,K)   : k  + (j*B+b)*ld_qkv + h*DH
    const bf16* v;
    const bf16* rd;     // [K][ld_rd] distance-indexed, + h*DH
    const float* u;     // r_w_bias [H][DH]
    const float* vb;    // r_r_bias [H][DH]
    const unsigned char* reset;   // [B] or null
    bf16* qu2;          // [T*B][H*DH] (q+u)*scale*log2e : written by fwd (may be null), read by bwd
    bf16* qv2;
    const bf16* dout;   // dO rows like out, ld_o
    const float* lse_in;
    const float* delta;
    const bf16* o_in;   // backward: forward output (or null); bwd_q then computes delta itself and writes it to `delta`
    bf16* out;          // [T*B][ld_o]
    float* lse;         // [B][H][T]
    bf16* dq;           // [T*B][H*DH] AC part of dq
    bf16* dk;           // rows like k, ld_dqkv
    bf16* dv;
    bf16* dsk;          // [H][T*B][ld_dsk]  dS indexed by distance d (zero-initialised by the caller)
    float* du_part;     // [B*QT][H*DH] column sums of dq (AC part)
    bf16* pbuf;         // P scratch [B*H][ceil(T/16)][ceil(K/64)][64 keys][16 rows]: written by bwd_q, read by bwd_kv2 (or null)
    // Probabilities saved by the FORWARD pass (relattn_fwd3 writes, relattn_bwd_q<.., FROMP> reads; null: the backward
    // recomputes its scores): pf = [B*H][ceil(T/32)][ceil(K/32)] tiles of PF_TILE_BYTES.  Bytes 0 .. 2047 of a tile are the
    // forward kernel's accumulator order -- lane (query i & 31, key half (jj >> 2) & 1) holds 16 bf16, register r = key
    // jj = 8 (r >> 2) + 4 half + 3 - (r & 3) of the 32-key sub-tile -- of e = exp2(s - m), m = the row's RUNNING maximum when
    // that sub-tile was exponentiated (log2 domain), sign bit = dropped by the attention dropout; bytes 2048 .. 2175 are
    // that m for the 32 queries (fp32).  P = |e| * exp2(m - lse * log2 e).
    void* pf;
    int p_layout;       // 0: the block order above; 1: [B*H][ceil(T/32)][2 ceil(K/64)] blocks of [32 keys][2 halves][4][4 queries],
                        //    the accumulator order of relattn_bwd_kv3_kernel; 2: the same block grid in the accumulator order of
                        //    relattn_bwd_q3_kernel (lane (query, half), registers 8k .. 8k+7 at 1024 k + 16 (query + 32 half))
    int ld_qkv, ld_rd, ld_o, ld_dqkv, ld_dsk;
    int dsk_wedge;      // > 0: dsk is uninitialised; zero columns i+M+1 .. i+M+dsk_wedge of every row (band GEMM contract)
    int dsk_tiled;      // != 0: dsk is stored as [H][T*B/64][ld_dsk/128] tiles of [64 rows][128 distances] (band.hip)
    int T, M, B, H;
    int same_length, sshift;
    float scale;
    unsigned drop_seed, drop_thr;   // attention-probability dropout: 16-bit threshold (0: off), see DropLane
    float drop_scale;
};

// relattn3.hip (d_head 64): forward on the 32x32 MFMA / transposed-score layout
int launch_relattn_fwd3(const AttnArgs& a, hipStream_t stream);
// relattn_kv3.hip (d_head 64): key-stationary backward from stored probabilities (p_layout 1) on the same MFMA
int launch_relattn_bwd_kv3(const AttnArgs& a, hipStream_t stream);
// relattn_q3.hip (d_head 64): query-stationary backward on the 32x32 MFMA / transposed-score layout; stores P in p_layout 2
bool relattn_bwd_q3_takes(const AttnArgs& a);
int launch_relattn_bwd_q3(const AttnArgs& a, hipStream_t stream);

constexpr int PF_TILE_BYTES = 2176;

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ bool is_masked(int i, int j, int M, int same_length, int sshift, bool rst) {
    return (j > i + M) || (same_length && j <= i - sshift) || (rst && j < M);
}
// kv tile range visible from query rows [i0, i0+qrows-1]
__device__ __forceinline__ void kv_range(const AttnArgs& a, int i0, int qrows, bool rst, int& jt_lo, int& jt_hi) {
    const int K = a.T + a.M;
    int jlo = rst ? a.M : 0;
    if (a.same_length) jlo = max(jlo, i0 - a.sshift + 1);
    jlo = max(jlo, 0);
    const int jhi = min(K - 1, i0 + qrows - 1 + a.M);
    jt_lo = jlo >> 6;
    jt_hi = jhi >> 6;
}

typedef __amdgpu_buffer_rsrc_t srd_t;
__device__ __forceinline__ srd_t make_srd(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 buf_ld(srd_t r, unsigned byte_off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

// Workgroups go to the 8 XCDs round-robin by linear id.  Hand every XCD whole (batch, head) pairs: all
// tiles of a pair share K, V and the band (they hit in that XCD's L2), and the heavy and light tiles of
// the causal triangle land on the same XCD, so the per-XCD work is balanced.  (A 3-D grid with the tile
// index fastest puts tile t on XCD t % 8: 2.4x more work on XCD 0 than on XCD 7 at 16 tiles.)
__device__ __forceinline__ void tile_coords(int ntile, int H, int B, int& tile, int& h, int& b) {
    const int id = blockIdx.x, NP = H * B;
    int pair;
    if ((NP & 7) == 0) {
        const int slot = id >> 3;
        pair = (slot / ntile) * 8 + (id & 7);
        tile = slot % ntile;
    } else {
        pair = id / ntile;
        tile = id % ntile;
    }
    b = pair / H;
    h = pair - b * H;
}

__device__ __forceinline__ void lds_dma16(srd_t srd, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(srd) : "memory");
}
// (tile offset in the scalar operand: one descriptor per stream of tiles)
__device__ __forceinline__ void lds_dma16s(srd_t srd, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

}  // namespace
