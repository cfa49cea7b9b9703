// Single-token decode step of the generation loop with a per-layer K/V cache (gfx950).
//
// Reference: InferenceTask.calc_logits_and_mems -> MemTransformerLM.forward_generate with qlen = 1
// (commu/midi_generator/midi_inferrer.py:199-207, commu/model/model.py:606-628).  The reference
// keeps the per-layer hidden states ("mems") and re-projects K and V of the WHOLE memory on every
// step (model.py:283-288); with fixed weights the projections are identical each time, so they are
// cached instead ([B][H][Lmax][DH] per layer, bf16) and a step reads each cached row
// once: HBM-bound.  Sequences are independent and RAGGED: klen[b] is the number of valid rows of
// sequence b; the new token has position klen[b] and attends to rows 0..klen[b] with distance
// d = klen[b] - j.  A step whose result the reference discards (quirk Q3) or a sequence that is not
// stepping simply does not advance klen[b].
//
//  kv_append : cache[b][klen[b]] = (k, v) of the new token (active sequences only)
//  decode_attn: one workgroup per (b, h): scores lane-per-key, softmax in LDS, P.V lane-per-feature
#include "common.h"
#include "commu_hip.h"

namespace {

// caches are head-major: [B][H][Lmax][DH], so the rows one workgroup streams are contiguous
__global__ void kv_append_kernel(const bf16* __restrict__ qkv, int ld_qkv, bf16* __restrict__ kc,
                                 bf16* __restrict__ vc, const int* __restrict__ klen,
                                 const unsigned char* __restrict__ active, int B, int Lmax, int H, int DH) {
    const int b = blockIdx.x;
    if (active != nullptr && !active[b]) return;
    const int pos = klen[b];
    if (pos >= Lmax) return;
    const int HD = H * DH;
    const bf16* src = qkv + (size_t)b * ld_qkv;
    for (int c = threadIdx.x * 8; c < HD; c += blockDim.x * 8) {
        const int h = c / DH, f = c - h * DH;
        const size_t off = (((size_t)b * H + h) * Lmax + pos) * DH + f;
        st_bf16x8(kc + off, ld_bf16x8(src + HD + c));
        st_bf16x8(vc + off, ld_bf16x8(src + 2 * HD + c));
    }
}

constexpr int DEC_MAXK = 4224;      // >= 4146 + 1 (memory_length of the inference config) rounded up

// sum over the 8 lanes that share (lane >> 3): xor 1, xor 2 (quad_perm), then mirror within 8
__device__ __forceinline__ float oct_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    return v;
}

// one workgroup per (b, h).  Both phases stream 16 bytes per lane: a wave instruction covers
// 64/LPR consecutive cache rows (LPR = DH/8 lanes per row), fully coalesced.
// UNR row-instructions per batch (measured at 1000 keys: 2 -> 33.5 us, 4 -> 28.5, 8 -> 28.8; the streaming part then runs at
// 5.9 TB/s and the remaining 6 us are launch + the dependent first loads)
template <int DH, int UNR = 4>
__global__ __launch_bounds__(256) void decode_attn_kernel(
    const bf16* __restrict__ qkv, int ld_qkv, const bf16* kc, const bf16* vc,
    const bf16* __restrict__ rd, int ld_rd, const float* __restrict__ u, const float* __restrict__ vb,
    const int* __restrict__ klen, const unsigned char* __restrict__ active, bf16* __restrict__ out, int ld_o,
    int H, int Lmax, float scale, int append, int nsplit, float* split_ws, unsigned* split_cnt) {
    constexpr int LPR = DH / 8;            // lanes per row (8 for DH 64, 4 for DH 32)
    constexpr int RPW = 64 / LPR;          // rows per wave instruction
    __shared__ float sS[DEC_MAXK];
    __shared__ float red[8];
    __shared__ float sO[4][RPW][DH];
    __shared__ int s_last;
    // nsplit > 1 (long memories, few live sequences: a (sequence, head) pair alone streams ~1 MB at 4000 keys): the keys of
    // a pair are split over up to nsplit workgroups (blockIdx = pair * nsplit + split); each leaves (max, sum, un-normalised
    // P.V) in split_ws, the LAST to arrive at the pair's counter combines them (write-through stores, drained, agent-scope
    // counter, write-through loads: MI355X_MICROARCH.md, inter-workgroup visibility) and resets the counter.
    const int pair = nsplit > 1 ? blockIdx.x / nsplit : blockIdx.x;
    const int split = nsplit > 1 ? blockIdx.x - pair * nsplit : 0;
    const int b = pair / H, h = pair - b * H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int sub = lane % LPR, rowl = lane / LPR;
    // one round trip for everything the step needs before the cache: length, active flag, q, the biases, and the new
    // token's K and V (every lane: the 8 features of its chunk)
    const int pos = klen[b];
    const unsigned char act = active != nullptr ? active[b] : (unsigned char)1;
    const bf16* qrow = qkv + (size_t)b * ld_qkv + h * DH + 8 * sub;
    const bf16x8 q8 = ld_bf16x8(qrow);
    const bf16x8 knew = ld_bf16x8(qrow + H * DH), vnew = ld_bf16x8(qrow + 2 * H * DH);
    float ub[8], vbb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ub[e] = u[h * DH + 8 * sub + e]; vbb[e] = vb[h * DH + 8 * sub + e]; }
    if (!act) return;
    // fused kv_append: this head's K and V of the new token go to cache row klen[b]; the step itself takes them from
    // registers (row `self`), so nothing waits for the store
    const int self = (append && pos < Lmax) ? pos : -1;
    if (self >= 0 && tid < 2 * LPR && split == 0) {
        bf16* dst = (bf16*)(tid < LPR ? kc : vc) + (((size_t)b * H + h) * Lmax + pos) * DH + 8 * sub;
        st_bf16x8(dst, tid < LPR ? knew : vnew);
    }
    const int nall = min(pos + 1, Lmax);           // keys 0..klen[b] (the new token included)
    // this workgroup's keys [jlo, n): chunks of >= 512 keys, a multiple of 64 (all of them without a split)
    int jlo = 0, n = nall, neff = 1;
    if (nsplit > 1) {
        int chunk = (nall + nsplit - 1) / nsplit;
        chunk = max(512, (chunk + 63) & ~63);
        neff = (nall + chunk - 1) / chunk;
        if (split >= neff) return;
        jlo = split * chunk;
        n = min(nall, jlo + chunk);
    }
    const bf16* kb = kc + ((size_t)b * H + h) * Lmax * DH + 8 * sub;
    const bf16* vbp = vc + ((size_t)b * H + h) * Lmax * DH + 8 * sub;
    const bf16* rb = rd + h * DH + 8 * sub;
    float qu[8], qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float q = bf2f(q8[e]);
        qu[e] = (q + ub[e]) * scale;
        qv[e] = (q + vbb[e]) * scale;
    }
    // ---- scores: RPW keys per wave instruction, UNR instructions' loads in flight together (the step is a single
    // pass over the cache: memory-level parallelism, not arithmetic, sets its speed)
    // Both loops are software-pipelined: the loads of batch i + 1 are issued before batch i is consumed, so a wave keeps
    // 2 UNR row-instructions (16 KB with the distance table) in flight; the first V batch is requested before the softmax.
    constexpr int STEP = 4 * RPW * UNR;
    float mx = -3.0e38f;
    {
        bf16x8 kk[UNR], r8[UNR], kn[UNR], rn[UNR];
        auto issue = [&](bf16x8 (&kd)[UNR], bf16x8 (&rdst)[UNR], int j0) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int jc = min(j0 + 4 * RPW * u + rowl, n - 1);
                kd[u] = ld_bf16x8(kb + (size_t)jc * DH);
                rdst[u] = ld_bf16x8(rb + (size_t)((nall - 1) - jc) * ld_rd);
            }
        };
        auto consume = [&](const bf16x8 (&kd)[UNR], const bf16x8 (&rdst)[UNR], int j0) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int j = j0 + 4 * RPW * u + rowl;
                float s = 0.f;
                const bf16x8 kx = (j == self) ? knew : kd[u];
#pragma unroll
                for (int e = 0; e < 8; ++e) s += qu[e] * bf2f(kx[e]) + qv[e] * bf2f(rdst[u][e]);
                s = (LPR == 8) ? oct_sum(s) : (s + dpp_f<0xB1>(s)) + dpp_f<0x4E>(s + dpp_f<0xB1>(s));
                if (j < n) {
                    if (sub == 0) sS[j - jlo] = s;
                    mx = fmaxf(mx, s);
                }
            }
        };
        // ping-pong (no register copies: a copy of a register that a load is still filling would wait for the load)
        int j0 = jlo + w * RPW;
        if (j0 < n) issue(kk, r8, j0);
        for (; j0 < n; j0 += 2 * STEP) {
            if (j0 + STEP < n) issue(kn, rn, j0 + STEP);
            consume(kk, r8, j0);
            if (j0 + 2 * STEP < n) issue(kk, r8, j0 + 2 * STEP);
            if (j0 + STEP < n) consume(kn, rn, j0 + STEP);
        }
    }
    bf16x8 v8[UNR], vn[UNR];
    auto issue_v = [&](bf16x8 (&vd)[UNR], int j0) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) vd[u] = ld_bf16x8(vbp + (size_t)min(j0 + 4 * RPW * u + rowl, n - 1) * DH);
    };
    if (jlo + w * RPW < n) issue_v(v8, jlo + w * RPW);
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < n - jlo; j += 256) {
        const float p = __expf(sS[j] - mx);
        sS[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + w] = sum;
    __syncthreads();
    const float lsum = red[4] + red[5] + red[6] + red[7];
    const float inv = 1.f / lsum;
    // ---- P.V: lane accumulates 8 features of the keys it visits
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto consume_v = [&](const bf16x8 (&vd)[UNR], int j0) {
        float p[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = j0 + 4 * RPW * u + rowl;
            p[u] = (j < n) ? sS[j - jlo] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const bf16x8 vx = (j0 + 4 * RPW * u + rowl == self) ? vnew : vd[u];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += p[u] * bf2f(vx[e]);
        }
    };
    for (int j0 = jlo + w * RPW; j0 < n; j0 += 2 * STEP) {
        if (j0 + STEP < n) issue_v(vn, j0 + STEP);
        consume_v(v8, j0);
        if (j0 + 2 * STEP < n) issue_v(v8, j0 + 2 * STEP);
        if (j0 + STEP < n) consume_v(vn, j0 + STEP);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sO[w][rowl][8 * sub + e] = acc[e];
    __syncthreads();
    float o = 0.f;
    if (tid < DH) {
        for (int ww = 0; ww < 4; ++ww)
            for (int r = 0; r < RPW; ++r) o += sO[ww][r][tid];
    }
    if (neff == 1) {
        if (tid < DH) out[(size_t)b * ld_o + h * DH + tid] = f2bf(o * inv);
        return;
    }
    // ---- split: publish (o[DH], max, sum), the last arriver of the pair combines
    constexpr int REC = DH + 2;
    float* rec = split_ws + ((size_t)pair * nsplit + split) * REC;
    if (tid < DH) __hip_atomic_store(rec + tid, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == DH) __hip_atomic_store(rec + DH, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == DH + 1) __hip_atomic_store(rec + DH + 1, lsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(split_cnt + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == (unsigned)(neff - 1);
        if (s_last) __hip_atomic_store(split_cnt + pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // next launch
    }
    __syncthreads();
    if (!s_last) return;
    if (tid < DH) {
        const float* base = split_ws + (size_t)pair * nsplit * REC;
        float m = -3.0e38f;
        for (int sidx = 0; sidx < neff; ++sidx)
            m = fmaxf(m, __hip_atomic_load(base + sidx * REC + DH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        float num = 0.f, den = 0.f;
        for (int sidx = 0; sidx < neff; ++sidx) {
            const float f = __expf(__hip_atomic_load(base + sidx * REC + DH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - m);
            num += f * __hip_atomic_load(base + sidx * REC + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            den += f * __hip_atomic_load(base + sidx * REC + DH + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        out[(size_t)b * ld_o + h * DH + tid] = f2bf(num / den);
    }
}

__global__ void klen_advance_kernel(int* __restrict__ klen, const unsigned char* __restrict__ advance, int B,
                                    int Lmax) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && advance[b] && klen[b] < Lmax - 1) klen[b] += 1;
}

}  // namespace

extern "C" int commu_decode_kv_append(const void* qkv, int ld_qkv, void* kcache, void* vcache, const int* klen,
                                      const unsigned char* active, int B, int Lmax, int H, int HD,
                                      hipStream_t stream) {
    if (B <= 0) return 0;
    if ((HD % 8) || (ld_qkv % 8) || H <= 0 || (HD % H) || ((HD / H) % 8)) return -22;
    COMMU_LAUNCH(kv_append_kernel, dim3(B), dim3(64), 0, stream, (const bf16*)qkv, ld_qkv, (bf16*)kcache,
                 (bf16*)vcache, klen, active, B, Lmax, H, HD / H);
    COMMU_LAUNCH_CHECK();
    return 0;
}

static int launch_decode_attn(const void* qkv, int ld_qkv, void* kcache, void* vcache, const void* rd, int ld_rd,
                              const float* r_w_bias, const float* r_r_bias, const int* klen, const unsigned char* active,
                              void* out, int ld_o, int B, int H, int DH, int Lmax, float scale, int append, int nsplit,
                              float* split_ws, unsigned* split_cnt, hipStream_t stream) {
    if (B <= 0) return 0;
    if (Lmax > DEC_MAXK || (ld_qkv % 8) || (ld_rd % 8)) return -22;
    if (nsplit < 1 || nsplit > 16 || (nsplit > 1 && (split_ws == nullptr || split_cnt == nullptr))) return -22;
    dim3 grid(B * H * nsplit);
    if (DH == 64)
        COMMU_LAUNCH(decode_attn_kernel<64>, grid, dim3(256), 0, stream, (const bf16*)qkv, ld_qkv,
                     (const bf16*)kcache, (const bf16*)vcache, (const bf16*)rd, ld_rd, r_w_bias, r_r_bias, klen,
                     active, (bf16*)out, ld_o, H, Lmax, scale, append, nsplit, split_ws, split_cnt);
    else if (DH == 32)
        COMMU_LAUNCH(decode_attn_kernel<32>, grid, dim3(256), 0, stream, (const bf16*)qkv, ld_qkv,
                     (const bf16*)kcache, (const bf16*)vcache, (const bf16*)rd, ld_rd, r_w_bias, r_r_bias, klen,
                     active, (bf16*)out, ld_o, H, Lmax, scale, append, nsplit, split_ws, split_cnt);
    else
        return -22;
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_decode_attn(const void* qkv, int ld_qkv, void* kcache, void* vcache,
                                 const void* rd, int ld_rd, const float* r_w_bias, const float* r_r_bias,
                                 const int* klen, const unsigned char* active, void* out, int ld_o, int B, int H,
                                 int DH, int Lmax, float scale, int append, hipStream_t stream) {
    return launch_decode_attn(qkv, ld_qkv, kcache, vcache, rd, ld_rd, r_w_bias, r_r_bias, klen, active, out, ld_o, B, H, DH,
                              Lmax, scale, append, 1, nullptr, nullptr, stream);
}

extern "C" int commu_decode_attn_split(const void* qkv, int ld_qkv, void* kcache, void* vcache,
                                       const void* rd, int ld_rd, const float* r_w_bias, const float* r_r_bias,
                                       const int* klen, const unsigned char* active, void* out, int ld_o, int B, int H,
                                       int DH, int Lmax, float scale, int append, int nsplit, float* split_ws,
                                       unsigned* split_cnt, hipStream_t stream) {
    return launch_decode_attn(qkv, ld_qkv, kcache, vcache, rd, ld_rd, r_w_bias, r_r_bias, klen, active, out, ld_o, B, H, DH,
                              Lmax, scale, append, nsplit, split_ws, split_cnt, stream);
}

extern "C" int commu_decode_advance(int* klen, const unsigned char* advance, int B, int Lmax, hipStream_t stream) {
    if (B <= 0) return 0;
    COMMU_LAUNCH(klen_advance_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, klen, advance, B, Lmax);
    COMMU_LAUNCH_CHECK();
    return 0;
}
