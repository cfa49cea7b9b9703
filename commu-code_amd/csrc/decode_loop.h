// Per-sequence stages of the decode loop iteration as device functions (one 64-lane wave per sequence), shared by the
// stand-alone kernels (sample.hip, forcing.hip) and the fused sample -> post -> pre launch (forcing.hip).
#pragma once
#include "common.h"

namespace {

// Wave-wide reductions and scans without LDS traffic: DPP inside the rows of 16 lanes, v_readlane across the four rows
// (a __shfl_* is a ds_bpermute_b32 + wait, ~130 cycles each; the sampling step made 54 of them in a row: 3 us).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ float lane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wsum_f(float v) {          // uniform result
    v = row16_sum(v);
    return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ __forceinline__ float wmax_f(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ int wmin_i(int v) {
    v = min(v, dpp_i<0xB1>(v)); v = min(v, dpp_i<0x4E>(v)); v = min(v, dpp_i<0x141>(v)); v = min(v, dpp_i<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wmax_i(int v) {
    v = max(v, dpp_i<0xB1>(v)); v = max(v, dpp_i<0x4E>(v)); v = max(v, dpp_i<0x141>(v)); v = max(v, dpp_i<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// inclusive prefix sums in lane order: row_shr 1, 2, 4, 8 (lanes without a source add 0), then the totals of the rows before
__device__ __forceinline__ float wscan_f(float v, int lane) {
    v += dpp_f<0x111>(v); v += dpp_f<0x112>(v); v += dpp_f<0x114>(v); v += dpp_f<0x118>(v);
    const float t0 = lane_f(v, 15), t1 = lane_f(v, 31), t2 = lane_f(v, 47);
    const int r = lane >> 4;
    return v + ((r >= 1 ? t0 : 0.f) + (r >= 2 ? t1 : 0.f) + (r >= 3 ? t2 : 0.f));
}
__device__ __forceinline__ int wscan_i(int v, int lane) {
    v += dpp_i<0x111>(v); v += dpp_i<0x112>(v); v += dpp_i<0x114>(v); v += dpp_i<0x118>(v);
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int r = lane >> 4;
    return v + (r >= 1 ? t0 : 0) + (r >= 2 ? t1 : 0) + (r >= 3 ? t2 : 0);
}

// ------------------------------------------------------------------------------------------------ sampling step (K15)
constexpr int PER_LANE = 12;      // 64 * 12 = 768 >= 729: lane l owns ids [12 l, 12 l + 12)

// one wave per sequence b (lane = threadIdx.x of a 64-thread workgroup)
__device__ __forceinline__ int sample_topk_body(int b, int lane, float* __restrict__ logits, int ld, int V,
                                                 const unsigned char* __restrict__ wrong, int ldw,
                                                 const float* __restrict__ uni,
                                                 const unsigned char* __restrict__ active, float temperature,
                                                 int top_k, int* __restrict__ token, float* __restrict__ probs_out,
                                                 int ldp, float top_p = 1.f) {
    const unsigned char act = active != nullptr ? active[b] : (unsigned char)1;          // (tested after the loads below are
    float* lg = logits + (size_t)b * ld;                                                  //  issued: one round trip, not two)
    float p[PER_LANE];
    const int base = lane * PER_LANE;
    // every load of the step up front and unconditional (clamped index): one memory round trip, not one per element
    float raw[PER_LANE];
#pragma unroll
    for (int e = 0; e < PER_LANE; ++e) raw[e] = lg[min(base + e, V - 1)];
    unsigned wmask = 0u;          // the rejected ("wrong") tokens of this lane
    if (wrong != nullptr) {
        unsigned char wb[PER_LANE];
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) wb[e] = wrong[(size_t)b * ldw + min(base + e, V - 1)];
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e)
            if (base + e < V && wb[e] != 0) wmask |= 1u << e;
    }
    const float u = uni != nullptr ? uni[b] : 0.5f;
    if (!act) return -2;
    // ---- calc_probs
    if (temperature == 0.f) {
        float best = -INFINITY;
        int bi = V;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) {
            const int id = base + e;
            if (id >= 1 && id < V && raw[e] > best) { best = raw[e]; bi = id; }
        }
        const float wbest = wmax_f(best);                      // ties: the lowest id
        bi = wmin_i(best == wbest ? bi : V);
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) p[e] = (base + e == bi) ? 1.f : 0.f;
    } else {
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) {
            const int id = base + e;
            const bool in = id >= 1 && id < V;
            const float x = in ? raw[e] / temperature : -INFINITY;
            if (in) lg[id] = x;                 // in-place division: compounds on a redo (Q5)
            p[e] = x;
            mx = fmaxf(mx, x);
        }
        mx = wmax_f(mx);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) {
            p[e] = (p[e] == -INFINITY) ? 0.f : expf(p[e] - mx);
            s += p[e];
        }
        s = wsum_f(s);
        const float inv = 1.f / s;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) p[e] *= inv;
    }
    // ---- apply_sampling: top-k (ties: lowest id first).  Probabilities are >= 0, so their bit patterns order like the
    // values: the k-th largest key is found by a radix select, most significant byte first -- per byte one 256-bin
    // histogram of the remaining candidates in LDS, then the bin in which the count from the top reaches k (4 bins per
    // lane, one DPP scan) -- four passes instead of the 32 bit-by-bit counting rounds of the previous version (4.8 us of
    // the step's 8.2); elements equal to the threshold are admitted in id order until k are kept.
    unsigned keep = 0u;
    {
        __shared__ int hist[256];
        unsigned key[PER_LANE];
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) key[e] = (base + e < V) ? __float_as_uint(p[e]) : 0u;
        unsigned thr = 0u, known = 0u;          // bytes of the threshold found so far / mask of those bytes
        int krem = top_k;                        // rank (from the largest) of the threshold among the remaining candidates
#pragma unroll 1
        for (int shift = 24; shift >= 0; shift -= 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) hist[lane * 4 + q] = 0;
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): one wave, LDS operations complete in order
#pragma unroll
            for (int e = 0; e < PER_LANE; ++e)
                if (base + e < V && ((key[e] ^ thr) & known) == 0u) atomicAdd(&hist[(key[e] >> shift) & 255u], 1);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            // lane l owns bins 255 - 4 l ... 252 - 4 l: lane order = descending bin order
            int h[4], tot = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { h[q] = hist[255 - 4 * lane - q]; tot += h[q]; }
            const int incl = wscan_i(tot, lane), before = incl - tot;
            const bool mine = before < krem && krem <= incl;          // exactly one lane
            int bin = 0, above = 0;
            if (mine) {
                int c = before;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (c < krem && krem <= c + h[q]) { bin = 255 - 4 * lane - q; above = c; }
                    c += h[q];
                }
            }
            const int src = __builtin_ctzll(__ballot(mine));
            bin = __builtin_amdgcn_readlane(bin, src);
            above = __builtin_amdgcn_readlane(above, src);
            thr |= (unsigned)bin << shift;
            known |= 255u << shift;
            krem -= above;
        }
        int ngt = 0, neq_lane = 0;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) {
            ngt += __popcll(__ballot(key[e] > thr));
            neq_lane += (key[e] == thr && base + e < V) ? 1 : 0;
        }
        // exclusive prefix of the per-lane tie counts in lane (= id) order
        const int incl = wscan_i(neq_lane, lane);
        int rank = incl - neq_lane;
        const int room = top_k - ngt;          // ties admitted
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) {
            if (key[e] > thr) keep |= 1u << e;
            else if (key[e] == thr && base + e < V) {
                if (rank < room) keep |= 1u << e;
                ++rank;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < PER_LANE; ++e) {
        const int id = base + e;
        const bool k = (((keep & ~wmask) >> e) & 1u) && id < V;
        p[e] = k ? p[e] : 0.f;
        s += p[e];
    }
    const float tot = wsum_f(s);
    if (!(tot > 0.f)) {                          // NaN / zero mass: the reference's multinomial raises
        if (lane == 0) token[b] = -1;
        if (probs_out != nullptr)
            for (int e = 0; e < PER_LANE; ++e)
                if (base + e < V) probs_out[(size_t)b * ldp + base + e] = NAN;
        return -1;
    }
    const float inv = 1.f / tot;
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < PER_LANE; ++e) { p[e] *= inv; ls += p[e]; }
    // ---- nucleus ("top-p") filter, an extra mode the reference does not have (top_p >= 1: off).  Applied to the
    // distribution left by the top-k / rejected-token step: in order of decreasing probability (ties: lowest id first) a
    // token is kept while the mass BEFORE it is < top_p; the survivors are renormalised.  The smallest kept probability
    // v* is the largest threshold t with mass{p >= t} >= top_p: the same bitwise search as the top-k threshold, on masses.
    if (top_p < 1.f) {
        unsigned key[PER_LANE];
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) key[e] = (base + e < V) ? __float_as_uint(p[e]) : 0u;
        unsigned thr = 0u;
        for (int bit = 30; bit >= 0; --bit) {          // (p <= 1: bit 31 is never set)
            const unsigned cand = thr | (1u << bit);
            float m = 0.f;
#pragma unroll
            for (int e = 0; e < PER_LANE; ++e) m += key[e] >= cand ? p[e] : 0.f;
            if (wsum_f(m) >= top_p) thr = cand;
        }
        float mg = 0.f;
        int neq_lane = 0;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) {
            mg += key[e] > thr ? p[e] : 0.f;
            neq_lane += (thr != 0u && key[e] == thr) ? 1 : 0;
        }
        mg = wsum_f(mg);
        const int incl_t = wscan_i(neq_lane, lane);
        const int nties = __builtin_amdgcn_readlane(incl_t, 63);
        int room = 0;          // ties admitted, in id order: the smallest r >= 1 with mg + r v* >= top_p
        if (thr != 0u) room = max(1, min(nties, (int)ceilf((top_p - mg) / __uint_as_float(thr))));
        int rank = incl_t - neq_lane;
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) {
            bool k = key[e] > thr;
            if (thr != 0u && key[e] == thr) { k = rank < room; ++rank; }
            p[e] = k ? p[e] : 0.f;
            s2 += p[e];
        }
        const float inv2 = 1.f / wsum_f(s2);
        ls = 0.f;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) { p[e] *= inv2; ls += p[e]; }
    }
    if (probs_out != nullptr)
        for (int e = 0; e < PER_LANE; ++e)
            if (base + e < V) probs_out[(size_t)b * ldp + base + e] = p[e];
    // ---- infer_token: smallest id with cdf[id] > u
    const float incl = wscan_f(ls, lane);
    const float excl = incl - ls;
    int cand = 1 << 30;
    float c = excl;
#pragma unroll
    for (int e = 0; e < PER_LANE; ++e) {
        c += p[e];
        if (p[e] > 0.f && c > u && cand == (1 << 30)) cand = base + e;
    }
    // fall-back for u above the accumulated total (rounding): the last token with mass
    int last = -1;
#pragma unroll
    for (int e = 0; e < PER_LANE; ++e)
        if (p[e] > 0.f) last = base + e;
    cand = wmin_i(cand);
    last = wmax_i(last);
    const int drawn = (cand == (1 << 30)) ? last : cand;
    if (lane == 0) token[b] = drawn;
    return drawn;
}


// ------------------------------------------------------------------------------------------------ forcing rules
constexpr int TOK_EOS = 1, TOK_BAR = 2, TOK_CHORD_LO = 195, TOK_CHORD_HI = 303, TOK_POS0 = 432, POS_RES = 128;
constexpr int VOCAB = 729;

// field indices of the int32 state record (commu_forcing_state_ints() ints per sequence)
enum {
    F_LEN = 0,      // tokens in seq
    F_FORCED,       // token to feed next iteration, -1: none  (next_tokens_forced holds at most one token)
    F_REDO,         // no_sequence_appended: draw again from the current logits
    F_FIRST,        // first model step after the context: its memory is discarded
    F_FILLED,       // incomplete_filled
    F_DONE,
    F_FAILED,       // nothing could be drawn (Q12)
    F_ITERS,
    F_NBAR,         // seq.count(BAR)
    F_NCHORD,       // chord_length
    F_CUR,          // chords consumed so far
    F_LENGTH_FIT,   // chord_length == int(num_measures // 4 * 4)
    F_NDRAW,        // uniform variates consumed
    F_NTRACE,       // model steps recorded
    F_COUNT
};

// The state record of a sequence as registers of lane 0: loaded with one batch of loads, stored back once (the record's
// fields are read and written many times by the transitions; through memory every access after a store is a round trip).
__device__ __forceinline__ void record_load(int (&r)[F_COUNT], const int* st, int b) {
#pragma unroll
    for (int f = 0; f < F_COUNT; ++f) r[f] = st[(size_t)b * F_COUNT + f];
}
__device__ __forceinline__ void record_store(const int (&r)[F_COUNT], int* st, int b) {
#pragma unroll
    for (int f = 0; f < F_COUNT; ++f) st[(size_t)b * F_COUNT + f] = r[f];
}

__device__ __forceinline__ void forcing_pre_body(int b, int lane, int (&s)[F_COUNT], int* seq, int ld_seq,
                                                 const int* __restrict__ chord_tok, const int* __restrict__ chord_pos,
                                                 int ld_chord, unsigned char* wrong, const float* __restrict__ utable,
                                                 int ld_u, int max_iters, long long* tok, unsigned char* active,
                                                 unsigned char* keep, unsigned char* draw, float* uni, int* trace,
                                                 int ld_trace) {
    int* sq = seq + (size_t)b * ld_seq;
    int clear = 0;
    if (lane == 0) {
        int act = 0, kp = 0, dr = 0;
        long long t = 0;
        const int len = s[F_LEN];
        const int last = sq[len - 1], prev = len >= 2 ? sq[len - 2] : -1;
        if (!s[F_DONE] && (s[F_ITERS] >= max_iters || last == TOK_EOS || len >= ld_seq)) s[F_DONE] = 1;
        if (!s[F_DONE]) {
            s[F_ITERS] += 1;
            if (s[F_FORCED] >= 0) {                                     // midi_inferrer.py:247-251
                const int f = s[F_FORCED];
                s[F_FORCED] = -1;
                sq[len] = f;
                s[F_LEN] = len + 1;
                if (f == TOK_BAR) s[F_NBAR] += 1;
                t = f; act = 1; kp = 1;
            } else {
                if (s[F_REDO]) {                                        // :253-255
                    s[F_REDO] = 0;
                } else if (s[F_FIRST]) {                                // :256-258
                    s[F_FIRST] = 0;
                    t = last; act = 1; kp = 0;
                } else {                                                // :259-260
                    t = last; act = 1; kp = 1;
                }
                if (!s[F_FILLED]) s[F_FILLED] = s[F_NBAR] > 1;          // :267-268
                const int cur = s[F_CUR];
                const bool remnant = cur < s[F_NCHORD];
                bool decided = false;
                if (s[F_FILLED] && last == TOK_BAR) {                   // :271-273
                    s[F_FORCED] = TOK_POS0;
                    decided = true;
                } else if (remnant && s[F_FILLED]) {                    // :276-283
                    const int cp = chord_pos[(size_t)b * ld_chord + cur];
                    const bool posfit = prev == TOK_BAR && last == TOK_POS0;
                    const bool due = s[F_LENGTH_FIT] ? posfit : (posfit || (last == cp && cp != TOK_POS0));
                    if (due) {
                        s[F_FORCED] = chord_tok[(size_t)b * ld_chord + cur];
                        s[F_CUR] = cur + 1;
                        clear = 1;
                        decided = true;
                    }
                }
                if (!decided) {
                    dr = 1;
                    const int nd = s[F_NDRAW];
                    uni[b] = utable[(size_t)b * ld_u + (nd < ld_u ? nd : ld_u - 1)];
                    s[F_NDRAW] = nd + 1;
                }
            }
            if (act && trace != nullptr) {
                const int nt = s[F_NTRACE];
                if (2 * nt + 1 < ld_trace) {
                    trace[(size_t)b * ld_trace + 2 * nt] = (int)t;
                    trace[(size_t)b * ld_trace + 2 * nt + 1] = kp;
                }
                s[F_NTRACE] = nt + 1;
            }
        }
        tok[b] = t;
        active[b] = (unsigned char)act;
        keep[b] = (unsigned char)kp;
        draw[b] = (unsigned char)dr;
    }
    clear = __builtin_amdgcn_readfirstlane(clear);
    if (clear)
        for (int i = lane; i < VOCAB; i += 64) wrong[(size_t)b * VOCAB + i] = 0;
}

// token_val: the token drawn this iteration when the caller has it in a register (>= -1), else (-3) it is read from token[b]
__device__ __forceinline__ void forcing_post_body(int b, int lane, int (&s)[F_COUNT], int* seq, int ld_seq,
                                                  const int* __restrict__ chord_pos, int ld_chord, unsigned char* wrong,
                                                  const unsigned char* draw, const int* token, int* live, int* klen,
                                                  const unsigned char* keep, int lmax, int token_val = -3) {
    int clear = 0;
    if (lane == 0) {
        // memory length of the step that just ran: it grows unless the step's memory is discarded (quirk Q3)
        if (klen != nullptr && keep[b] && klen[b] < lmax - 1) klen[b] += 1;
        if (draw[b]) {
            const int t = token_val >= -1 ? token_val : token[b];
            const int cur = s[F_CUR];
            const bool remnant = cur < s[F_NCHORD];
            const int cp = remnant ? chord_pos[(size_t)b * ld_chord + cur] : -1;
            const bool inter = remnant && cp != TOK_POS0;
            if (t < 0) {                                                                  // :286-291, Q12
                s[F_FAILED] = 1;
                s[F_DONE] = 1;
            } else if (inter && ((cp < t && t < TOK_POS0 + POS_RES) || t == TOK_BAR)) {    // :294-296
                s[F_FORCED] = cp;
                clear = 1;
            } else if (t >= TOK_CHORD_LO && t <= TOK_CHORD_HI) {                           // :299-301
                s[F_REDO] = 1;
                wrong[(size_t)b * VOCAB + t] = 1;
            } else if (remnant && t == TOK_EOS) {                                         // :304-306
                s[F_FORCED] = inter ? cp : TOK_BAR;
            } else if (!remnant && t == TOK_BAR) {                                        // :309-311
                s[F_FORCED] = TOK_EOS;
            } else {
                const int len = s[F_LEN];
                if (len < ld_seq) {
                    seq[(size_t)b * ld_seq + len] = t;
                    s[F_LEN] = len + 1;
                }
                if (t == TOK_BAR) s[F_NBAR] += 1;
            }
        }
        if (live != nullptr && !s[F_DONE]) atomicAdd(live, 1);
    }
    clear = __builtin_amdgcn_readfirstlane(clear);
    if (clear)
        for (int i = lane; i < VOCAB; i += 64) wrong[(size_t)b * VOCAB + i] = 0;
}


}  // namespace
