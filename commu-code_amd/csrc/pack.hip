// Host-side batch packer of the training iterator (no device code): one call assembles a [T, B] batch from the flat
// corpus -- what the reference does with a Python loop over every column of every batch
// (commu/model/dataset.py:139-170: data[:n, i] = seq[pos : pos + n], target = seq[pos + 1 : pos + n + 1], pad
// elsewhere).  Called through ctypes from the prefetch thread: the call releases the GIL, so packing runs beside the
// training loop's Python instead of time-slicing with it (the numpy form needed the GIL ~15 times per batch and the
// training loop waited ~8 ms per step for it).
#include "commu_hip.h"
#include <stdint.h>

extern "C" long long commu_pack_batch(const int64_t* tokens, const int64_t* offsets, const int64_t* seq,
                                      const int64_t* pos, const int64_t* cnt, int B, int T, int64_t pad, int64_t* data,
                                      int64_t* target) {
    long long total = 0;
    if (B <= 0 || T <= 0) return -22;
    // row-major [T][B] outputs: walk rows outermost so that both output streams are written sequentially; columns in
    // blocks of 1024 (the per-column cursors live on the stack)
    for (int c0 = 0; c0 < B; c0 += 1024) {
        const int nb = B - c0 < 1024 ? B - c0 : 1024;
        const int64_t* base[1024];
        int64_t n[1024];
        for (int c = 0; c < nb; ++c) {
            const bool live = seq[c0 + c] >= 0 && cnt[c0 + c] > 0;
            base[c] = live ? tokens + offsets[seq[c0 + c]] + pos[c0 + c] : nullptr;
            n[c] = live ? cnt[c0 + c] : 0;
            total += n[c];
        }
        for (int r = 0; r < T; ++r) {
            int64_t* d = data + (int64_t)r * B + c0;
            int64_t* t = target + (int64_t)r * B + c0;
            for (int c = 0; c < nb; ++c) {
                const bool in = r < n[c];
                d[c] = in ? base[c][r] : pad;
                t[c] = in ? base[c][r + 1] : pad;
            }
        }
    }
    return total;
}
