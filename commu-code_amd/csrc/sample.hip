// K15: the sampling step of the decode loop, one wave per sequence, no host round trip inside.
// Reference: commu/midi_generator/midi_inferrer.py:209-237
//   calc_probs    : logits[1:] /= temperature (IN PLACE, quirk Q5), softmax, left-pad 0 (Q6)
//                   (temperature == 0: one-hot at argmax)
//   apply_sampling: keep the top-k probabilities, zero the rejected ("wrong") tokens, renormalise
//   infer_token   : draw one token from the result -- here by inverse CDF with an injected
//                   uniform variate u[b] (torch.multinomial's stream is not reproducible).
// A sequence whose kept mass is 0 (Q12: greedy argmax is a rejected token) gets token -1.
#include "common.cuh"
#include "commu_hip.h"

namespace {

constexpr int PER_LANE = 12;      // 64 * 12 = 768 >= 729: lane l owns ids [12 l, 12 l + 12)

__global__ __launch_bounds__(64) void sample_topk_kernel(float* __restrict__ logits, int ld, int V,
                                                         const unsigned char* __restrict__ wrong, int ldw,
                                                         const float* __restrict__ uni,
                                                         const unsigned char* __restrict__ active,
                                                         float temperature, int top_k,
                                                         int* __restrict__ token, float* __restrict__ probs_out,
                                                         int ldp) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (active != nullptr && !active[b]) return;
    float* lg = logits + (size_t)b * ld;
    float p[PER_LANE];
    const int base = lane * PER_LANE;
    // ---- calc_probs
    if (temperature == 0.f) {
        float best = -INFINITY;
        int bi = V;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) {
            const int id = base + e;
            if (id >= 1 && id < V && lg[id] > best) { best = lg[id]; bi = id; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) p[e] = (base + e == bi) ? 1.f : 0.f;
    } else {
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) {
            const int id = base + e;
            float x = -INFINITY;
            if (id >= 1 && id < V) {
                x = lg[id] / temperature;
                lg[id] = x;                     // in-place division: compounds on a redo (Q5)
            }
            p[e] = x;
            mx = fmaxf(mx, x);
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) {
            p[e] = (p[e] == -INFINITY) ? 0.f : expf(p[e] - mx);
            s += p[e];
        }
        s = wave_sum(s);
        const float inv = 1.f / s;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) p[e] *= inv;
    }
    // ---- apply_sampling: top-k (ties: lowest id first).  Probabilities are >= 0, so their bit patterns order like the
    // values: the k-th largest is found by a 32-step radix select whose counts are wave ballots + scalar popcounts (no
    // cross-lane data movement, ~1 us), instead of k rounds of a wave arg-max (12 LDS / DPP exchanges each: ~19 us at
    // k = 32); elements equal to the threshold are admitted in id order until k are kept.
    unsigned keep = 0u, wmask = 0u;          // wmask: the rejected ("wrong") tokens of this lane, read before the selection
    if (wrong != nullptr) {
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e)
            if (base + e < V && wrong[(size_t)b * ldw + base + e] != 0) wmask |= 1u << e;
    }
    {
        unsigned key[PER_LANE];
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) key[e] = (base + e < V) ? __float_as_uint(p[e]) : 0u;
        unsigned thr = 0u;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = thr | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int e = 0; e < PER_LANE; ++e) cnt += __popcll(__ballot(key[e] >= cand));
            if (cnt >= top_k) thr = cand;
        }
        int ngt = 0, neq_lane = 0;
#pragma unroll
        for (int e = 0; e < PER_LANE; ++e) {
            ngt += __popcll(__ballot(key[e] > thr));
            neq_lane += (key[e] == thr && base + e < V) ? 1 : 0;
        }
        // exclusive prefix of the per-lane tie counts in lane (= id) order
        int incl = neq_lane;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
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
    const float tot = wave_sum(s);
    if (!(tot > 0.f)) {                          // NaN / zero mass: the reference's multinomial raises
        if (lane == 0) token[b] = -1;
        if (probs_out != nullptr)
            for (int e = 0; e < PER_LANE; ++e)
                if (base + e < V) probs_out[(size_t)b * ldp + base + e] = NAN;
        return;
    }
    const float inv = 1.f / tot;
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < PER_LANE; ++e) { p[e] *= inv; ls += p[e]; }
    if (probs_out != nullptr)
        for (int e = 0; e < PER_LANE; ++e)
            if (base + e < V) probs_out[(size_t)b * ldp + base + e] = p[e];
    // ---- infer_token: smallest id with cdf[id] > u
    float incl = ls;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const float excl = incl - ls;
    const float u = uni != nullptr ? uni[b] : 0.5f;
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cand = min(cand, __shfl_xor(cand, o, 64));
        last = max(last, __shfl_xor(last, o, 64));
    }
    if (lane == 0) token[b] = (cand == (1 << 30)) ? last : cand;
}

}  // namespace

extern "C" int commu_sample_topk(float* logits, int ld, int nseq, int V, const unsigned char* wrong, int ldw,
                                 const float* uniforms, const unsigned char* active, float temperature,
                                 int top_k, int* token, float* probs_out, int ldp, hipStream_t stream) {
    if (nseq <= 0) return 0;
    if (V > 64 * PER_LANE || top_k < 1 || top_k > V) return -22;
    COMMU_LAUNCH(sample_topk_kernel, dim3(nseq), dim3(64), 0, stream, logits, ld, V, wrong, ldw, uniforms,
                 active, temperature, top_k, token, probs_out, ldp);
    COMMU_LAUNCH_CHECK();
    return 0;
}
