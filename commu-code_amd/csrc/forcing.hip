// Device-resident chord / bar forcing of the decode loop (gfx950): the per-sequence control flow of
// InferenceTask.generate_sequence + TeacherForceTask (commu/midi_generator/midi_inferrer.py:239-320 and :16-144),
// so that one loop iteration of ALL sequences -- decide, model step, sampling step, book-keeping -- is a chain of
// kernels (captured in a hipGraph by commu_amd/generate.py) with no host round trip.
//
// The reference's rules are held as a per-sequence state record and two transition functions:
//   commu_forcing_pre  : before the model step.   STOP (EOS / iteration cap) | FEED a forced token | then which
//                        model step this iteration takes -- none (re-draw from the logits already divided by the
//                        temperature, quirk Q5), first step whose memory is discarded (Q3), normal step on the last
//                        token (re-feeding the last forced token, Q4) -- and whether a token is drawn or the next
//                        token is forced (position 432 after a bar; the next chord of the progression).
//   commu_forcing_post : after the draw.   position past a pending chord -> force that position | chord token
//                        drawn -> reject, remember it, redo | EOS with chords left -> force position / bar | bar
//                        with no chords left -> force EOS | else append.
// One 64-lane wave per sequence: lane 0 runs the transition, the wave clears the 729-entry rejected-token bitmap.
#include "decode_loop.h"
#include "commu_hip.h"

namespace {

__global__ __launch_bounds__(64) void forcing_pre_kernel(int* st, int* seq, int ld_seq, const int* __restrict__ chord_tok,
                                                         const int* __restrict__ chord_pos, int ld_chord,
                                                         unsigned char* wrong, const float* __restrict__ utable, int ld_u,
                                                         int max_iters, long long* tok, unsigned char* active,
                                                         unsigned char* keep, unsigned char* draw, float* uni, int* trace,
                                                         int ld_trace) {
    int rec[F_COUNT];
    if (threadIdx.x == 0) record_load(rec, st, blockIdx.x);
    forcing_pre_body(blockIdx.x, threadIdx.x, rec, seq, ld_seq, chord_tok, chord_pos, ld_chord, wrong, utable, ld_u, max_iters,
                     tok, active, keep, draw, uni, trace, ld_trace);
    if (threadIdx.x == 0) record_store(rec, st, blockIdx.x);
}

__global__ __launch_bounds__(64) void forcing_post_kernel(int* st, int* seq, int ld_seq, const int* __restrict__ chord_pos,
                                                          int ld_chord, unsigned char* wrong, const unsigned char* draw,
                                                          const int* token, int* live, int* klen, const unsigned char* keep,
                                                          int lmax) {
    int rec[F_COUNT];
    if (threadIdx.x == 0) record_load(rec, st, blockIdx.x);
    forcing_post_body(blockIdx.x, threadIdx.x, rec, seq, ld_seq, chord_pos, ld_chord, wrong, draw, token, live, klen, keep,
                      lmax);
    if (threadIdx.x == 0) record_store(rec, st, blockIdx.x);
}

// One launch for the three per-sequence stages that follow the model step: sampling step -> book-keeping (post) -> the
// decision of the NEXT iteration (pre).  A sequence's stages only exchange data of that sequence, and one wave runs
// them in order; the workgroup barriers order the wave's own global stores and loads between stages.
struct LoopStageArgs {
    float* logits; int ld, V;
    unsigned char* wrong;
    float temperature; int top_k; float top_p;
    int* token; float* probs_out; int ldp;
    int *st, *seq; int ld_seq;
    const int *chord_tok, *chord_pos; int ld_chord;
    const float* utable; int ld_u, max_iters;
    long long* tok;
    unsigned char *active, *keep, *draw;
    float* uni;
    int* step_trace; int ld_trace;
    int* klen; int lmax;
    unsigned long long* trace;      // (diagnostics) [sequence][4] 100 MHz timestamps: start, sampled, post done, pre done
};
__global__ __launch_bounds__(64) void sample_post_pre_kernel(LoopStageArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (a.trace != nullptr && lane == 0) a.trace[b * 4 + 0] = wall_clock64();
    int rec[F_COUNT];                                  // the record travels through the three stages in lane 0's registers;
    if (lane == 0) record_load(rec, a.st, b);          // its loads are in flight under the sampling step
    const int drawn = sample_topk_body(b, lane, a.logits, a.ld, a.V, a.wrong, VOCAB, a.uni, a.draw, a.temperature, a.top_k,
                                       a.token, a.probs_out, a.ldp, a.top_p);
    __syncthreads();
    if (a.trace != nullptr && lane == 0) a.trace[b * 4 + 1] = wall_clock64();
    forcing_post_body(b, lane, rec, a.seq, a.ld_seq, a.chord_pos, a.ld_chord, a.wrong, a.draw, a.token, nullptr, a.klen,
                      a.keep, a.lmax, drawn >= -1 ? drawn : -3);
    __syncthreads();
    if (a.trace != nullptr && lane == 0) a.trace[b * 4 + 2] = wall_clock64();
    forcing_pre_body(b, lane, rec, a.seq, a.ld_seq, a.chord_tok, a.chord_pos, a.ld_chord, a.wrong, a.utable, a.ld_u,
                     a.max_iters, a.tok, a.active, a.keep, a.draw, a.uni, a.step_trace, a.ld_trace);
    if (lane == 0) record_store(rec, a.st, b);
    if (a.trace != nullptr && lane == 0) a.trace[b * 4 + 3] = wall_clock64();
}

// dst[b][0:n] = src[b][0:n] for rows with mask[b] != 0 (the logits of the sequences that stepped: the others keep
// theirs for a possible re-draw, quirk Q5)
__global__ void copy_rows_masked_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds_,
                                        const unsigned char* __restrict__ mask, int n) {
    const int b = blockIdx.x;
    if (!mask[b]) return;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[(size_t)b * ldd + i] = src[(size_t)b * lds_ + i];
}

}  // namespace

extern "C" int commu_forcing_state_ints(void) { return F_COUNT; }

extern "C" int commu_forcing_pre(int* state, int* seq, int ld_seq, const int* chord_tok, const int* chord_pos,
                                 int ld_chord, unsigned char* wrong, const float* utable, int ld_u, int max_iters,
                                 long long* tok, unsigned char* active, unsigned char* keep, unsigned char* draw,
                                 float* uni, int* trace, int ld_trace, int B, hipStream_t stream) {
    if (B <= 0) return 0;
    if (ld_seq < 2 || ld_chord < 1 || ld_u < 1) return -22;
    COMMU_LAUNCH(forcing_pre_kernel, dim3(B), dim3(64), 0, stream, state, seq, ld_seq, chord_tok, chord_pos, ld_chord,
                 wrong, utable, ld_u, max_iters, tok, active, keep, draw, uni, trace, ld_trace);
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_forcing_post(int* state, int* seq, int ld_seq, const int* chord_pos, int ld_chord,
                                  unsigned char* wrong, const unsigned char* draw, const int* token, int* live,
                                  int* klen, const unsigned char* keep, int lmax, int B, hipStream_t stream) {
    if (B <= 0) return 0;
    COMMU_LAUNCH(forcing_post_kernel, dim3(B), dim3(64), 0, stream, state, seq, ld_seq, chord_pos, ld_chord, wrong,
                 draw, token, live, klen, keep, lmax);
    COMMU_LAUNCH_CHECK();
    return 0;
}

extern "C" int commu_copy_rows_masked_f32(float* dst, int ldd, const float* src, int lds_, const unsigned char* mask,
                                          int rows, int n, hipStream_t stream) {
    if (rows <= 0 || n <= 0) return 0;
    COMMU_LAUNCH(copy_rows_masked_kernel, dim3(rows), dim3(256), 0, stream, dst, ldd, src, lds_, mask, n);
    COMMU_LAUNCH_CHECK();
    return 0;
}

static unsigned long long* g_loop_trace = nullptr;
/* diagnostics: following commu_decode_sample_post_pre launches write buf[sequence][4] timestamps (null: off) */
extern "C" int commu_decode_loop_trace(unsigned long long* buf) {
    g_loop_trace = buf;
    return 0;
}

extern "C" int commu_decode_sample_post_pre(float* logits, int ld, int V, unsigned char* wrong, float temperature, int top_k,
                                            float top_p, int* token, float* probs_out, int ldp, int* state, int* seq, int ld_seq,
                                            const int* chord_tok, const int* chord_pos, int ld_chord, const float* utable,
                                            int ld_u, int max_iters, long long* tok, unsigned char* active,
                                            unsigned char* keep, unsigned char* draw, float* uni, int* trace, int ld_trace,
                                            int* klen, int lmax, int B, hipStream_t stream) {
    if (B <= 0) return 0;
    if (V != VOCAB || V > 64 * PER_LANE || top_k < 1 || top_k > V || ld_seq < 2 || ld_chord < 1 || ld_u < 1 || !(top_p > 0.f)) return -22;
    LoopStageArgs a{logits, ld, V, wrong, temperature, top_k, top_p, token, probs_out, ldp, state, seq, ld_seq, chord_tok, chord_pos,
                    ld_chord, utable, ld_u, max_iters, tok, active, keep, draw, uni, trace, ld_trace, klen, lmax, g_loop_trace};
    COMMU_LAUNCH(sample_post_pre_kernel, dim3(B), dim3(64), 0, stream, a);
    COMMU_LAUNCH_CHECK();
    return 0;
}
