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
#include "common.cuh"
#include "commu_hip.h"

namespace {

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

__global__ __launch_bounds__(64) void forcing_pre_kernel(int* __restrict__ st, int* __restrict__ seq, int ld_seq,
                                                         const int* __restrict__ chord_tok,
                                                         const int* __restrict__ chord_pos, int ld_chord,
                                                         unsigned char* __restrict__ wrong,
                                                         const float* __restrict__ utable, int ld_u, int max_iters,
                                                         long long* __restrict__ tok, unsigned char* __restrict__ active,
                                                         unsigned char* __restrict__ keep, unsigned char* __restrict__ draw,
                                                         float* __restrict__ uni, int* __restrict__ trace, int ld_trace) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int* s = st + (size_t)b * F_COUNT;
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
    clear = __shfl(clear, 0, 64);
    if (clear)
        for (int i = lane; i < VOCAB; i += 64) wrong[(size_t)b * VOCAB + i] = 0;
}

__global__ __launch_bounds__(64) void forcing_post_kernel(int* __restrict__ st, int* __restrict__ seq, int ld_seq,
                                                          const int* __restrict__ chord_pos, int ld_chord,
                                                          unsigned char* __restrict__ wrong,
                                                          const unsigned char* __restrict__ draw,
                                                          const int* __restrict__ token, int* __restrict__ live,
                                                          int* __restrict__ klen, const unsigned char* __restrict__ keep,
                                                          int lmax) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int* s = st + (size_t)b * F_COUNT;
    int clear = 0;
    if (lane == 0) {
        // memory length of the step that just ran: it grows unless the step's memory is discarded (quirk Q3)
        if (klen != nullptr && keep[b] && klen[b] < lmax - 1) klen[b] += 1;
        if (draw[b]) {
            const int t = token[b];
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
    clear = __shfl(clear, 0, 64);
    if (clear)
        for (int i = lane; i < VOCAB; i += 64) wrong[(size_t)b * VOCAB + i] = 0;
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
