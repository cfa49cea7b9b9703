// K15: the sampling step of the decode loop, one wave per sequence, no host round trip inside.
// Reference: commu/midi_generator/midi_inferrer.py:209-237
//   calc_probs    : logits[1:] /= temperature (IN PLACE, quirk Q5), softmax, left-pad 0 (Q6)
//                   (temperature == 0: one-hot at argmax)
//   apply_sampling: keep the top-k probabilities, zero the rejected ("wrong") tokens, renormalise
//   infer_token   : draw one token from the result -- here by inverse CDF with an injected
//                   uniform variate u[b] (torch.multinomial's stream is not reproducible).
// A sequence whose kept mass is 0 (Q12: greedy argmax is a rejected token) gets token -1.
#include "decode_loop.h"
#include "commu_hip.h"

namespace {

__global__ __launch_bounds__(64) void sample_topk_kernel(float* __restrict__ logits, int ld, int V,
                                                         const unsigned char* __restrict__ wrong, int ldw,
                                                         const float* __restrict__ uni,
                                                         const unsigned char* __restrict__ active,
                                                         float temperature, int top_k,
                                                         int* __restrict__ token, float* __restrict__ probs_out,
                                                         int ldp, float top_p) {
    sample_topk_body(blockIdx.x, threadIdx.x, logits, ld, V, wrong, ldw, uni, active, temperature, top_k, token, probs_out,
                     ldp, top_p);
}

}  // namespace

extern "C" int commu_sample_topk_topp(float* logits, int ld, int nseq, int V, const unsigned char* wrong, int ldw,
                                      const float* uniforms, const unsigned char* active, float temperature,
                                      int top_k, float top_p, int* token, float* probs_out, int ldp,
                                      hipStream_t stream) {
    if (nseq <= 0) return 0;
    if (V > 64 * PER_LANE || top_k < 1 || top_k > V || !(top_p > 0.f)) return -22;
    COMMU_LAUNCH(sample_topk_kernel, dim3(nseq), dim3(64), 0, stream, logits, ld, V, wrong, ldw, uniforms,
                 active, temperature, top_k, token, probs_out, ldp, top_p);
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_sample_topk(float* logits, int ld, int nseq, int V, const unsigned char* wrong, int ldw,
                                 const float* uniforms, const unsigned char* active, float temperature,
                                 int top_k, int* token, float* probs_out, int ldp, hipStream_t stream) {
    return commu_sample_topk_topp(logits, ld, nseq, V, wrong, ldw, uniforms, active, temperature, top_k, 1.f, token,
                                  probs_out, ldp, stream);
}
