/* commu_hip.h -- C ABI of libcommu_hip.so: MI355X (gfx950) kernels for ComMU's Transformer-XL
 * training + sampling hot path.
 *
 * The reference (POZAlabs/ComMU-code) is pure Python on PyTorch: it has no FFI layer.  Each
 * entry point below replaces the PyTorch op sequence at the cited reference lines
 * (paths relative to the reference repository root) and is what a binding for that path
 * would call; the ctypes binding a maintainer would add is shown in INTEGRATION.md and
 * shipped as commu-code_amd/commu_amd/_lib.py.
 *
 * Conventions: plain device pointers + sizes, no framework types; every call enqueues work on
 * `stream` and returns 0, a positive hipError_t, or -22 (EINVAL) for an unsupported shape; no
 * call allocates, synchronises or throws.  "bf16" buffers are `void*` to 16-bit bfloat16.
 * Activation rows are time-major, row m = t * B + b, exactly the reference's [T, B, *] layout.
 */
#ifndef COMMU_HIP_H
#define COMMU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- GEMM (nn.Linear forward/backward: model.py:205,212,278,164,167,46; autograd of the same) */
enum {
    COMMU_EPI_BIAS = 1,      /* C += bias[n]                    (Linear bias, model.py:164,167,40) */
    COMMU_EPI_RELU = 2,      /* C = max(C, 0)                   (nn.ReLU, model.py:165)            */
    COMMU_EPI_RESID = 4,     /* C += resid[m, n] (bf16)         (w + attn_out / inp + core_out)    */
    COMMU_EPI_RELUMASK = 8,  /* C = relu_mask[m,n] > 0 ? C : 0  (ReLU backward)                    */
    COMMU_EPI_OUT_F32 = 16,  /* C is fp32 (default bf16)                                           */
    COMMU_EPI_DROPOUT = 32,  /* C = keep(drop_seed, m*N+n) ? C/(1-p) : 0  (nn.Dropout, model.py:166,168,210) */
    /* ReLU backward from ONE BIT per element instead of the bf16 activations (1/16 of the bytes).  SIGNBITS_OUT: the
     * `relu_mask` argument is an OUTPUT of commu_gemm_nt_signbits_words(M, N, K) 32-bit words receiving (C > 0) after the
     * epilogue, in a layout private to the kernel; RELUBITS: `relu_mask` is such a buffer written by a GEMM of the same
     * M x N, C = bit ? C * mask_scale : 0.  Only shapes with a non-zero word count; no RESID / RELUMASK / OUT_F32. */
    COMMU_EPI_SIGNBITS_OUT = 64,
    COMMU_EPI_RELUBITS = 128
};
/* words of the sign-bit buffer for an M x N output (0: this shape / K does not take the flags above) */
long long commu_gemm_nt_signbits_words(int M, int N, int K, int lda, int ldb, int ldc);
/* C[M,N] = A[M,K] . B[N,K]^T with fused epilogue, applied in this order: bias, relu, dropout, resid,
 * relu-mask (kept values times mask_scale).  K % 32 == 0, lda/ldb % 8 == 0, ldc % 4 == 0. */
int commu_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N,
                       int K, const float* bias, const void* resid, int ldr, const void* relu_mask,
                       int ldm, int flags, unsigned drop_seed, float drop_p, float mask_scale,
                       hipStream_t stream);
/* Decode-step form (M <= 64 rows, K % 128 == 0, K <= 1024) with the PRECEDING LayerNorm fused in (model.py:179,352
 * followed by the Linear of :164 / :205 / :46): C = LN(z)[:, :K] . B^T with epilogue flags BIAS, RELU, RESID, OUT_F32;
 * LN over the first D columns of z (gamma, beta, eps; columns D..K-1 are zero padding); a_out (optional, bf16
 * [M][lda_out]) receives LN(z), which the next residual connection adds.  Other shapes: -22. */
int commu_gemm_nt_ln_bf16(const void* z, int ldz, const float* gamma, const float* beta, int D, float eps,
                          void* a_out, int lda_out, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                          const float* bias, const void* resid, int ldr, int flags, hipStream_t stream);
/* batched form: entry z uses A + z*strideA, B + z*strideB, C + z*strideC, resid + z*strideR (element
 * strides); a 64-wide tile is used when N <= 64 (per-head GEMMs).  Epilogue flags: RESID, OUT_F32.
 * Causal band (tri_B > 0): A is dS by distance, row m = i*tri_B + b is zero (or unwritten) beyond column
 * i + tri_M, so a row tile only contracts columns < i_max + tri_M + 1 (rounded up to 64): the caller
 * guarantees ZEROS in each row from column i + tri_M + 1 up to that limit (commu_attn_bwd_desc.dsk_wedge). */
int commu_gemm_nt_bf16_batched(const void* A, int lda, long long strideA, const void* B, int ldb,
                               long long strideB, void* C, int ldc, long long strideC, int M, int N, int K,
                               const void* resid, int ldr, long long strideR, int flags, int batch,
                               int tri_B, int tri_M, hipStream_t stream);
/* slabs[s][n,k] = sum_{m in slice s} A[m,n] * B[m,k]  (weight gradients dW = dY^T X).
 * mode 1: LDS transpose reads (ds_read_b64_tr_b16); mode 0: 16-bit gathers. */
int commu_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, float* slabs, int ldc,
                       size_t slab_stride, int M, int N, int K, int nslices, int mode, hipStream_t stream);
/* recommended number of m-slices for the tile the TN kernel picks at this shape (fills the 256 CUs) */
int commu_gemm_tn_slices(int M, int N, int K);
/* batched form: slabs[(z*nslices + s)][n,k] for batch entry z.  Causal band (tri_B > 0): column n of A is
 * zero (or unwritten) in rows m < (n - tri_M) * tri_B, so an output tile starting at n0 skips those rows. */
int commu_gemm_tn_bf16_batched(const void* A, int lda, long long strideA, const void* B, int ldb,
                               long long strideB, float* slabs, int ldc, size_t slab_stride, int M, int N, int K,
                               int nslices, int batch, int tri_B, int tri_M, hipStream_t stream);
/* Grouped form for the weight gradients of a layer (autograd of the nn.Linear calls at model.py:163-169,205,212):
 * up to 8 problems out_p[N_p, K_p] = A_p[M, N_p]^T . B_p[M, K_p] sharing the token count M, ONE launch of the
 * 256 x 256 x 64 eight-phase kernel over (token slice, output tile) workgroups; slice s of problem p lands at
 * slabs[s * slab_stride + out_off_p + n * K_p + k] and is summed by commu_reduce_slabs_f32.
 * N_p, K_p >= 128 and % 8 == 0, lda/ldb % 8 == 0 and >= 128, M >= 4096 (else -22: use commu_gemm_tn_bf16). */
typedef struct commu_tn_problem {
    const void* A;      /* bf16 [M][lda]  (dY)          */
    const void* B;      /* bf16 [M][ldb]  (layer input) */
    int lda, ldb, N, K;
    long long out_off;  /* element offset of this problem's [N, K] block inside a slab */
    long long colsum_off; /* >= 0: the slice's column sums of A (sum_m A[m, n]: the bias gradient of the Linear whose dY this
                           * is, commu/model/model.py:163-169) go to slabs[s * slab_stride + colsum_off + n], n < N -- one
                           * extra MFMA per 64 tokens in the workgroups of the first K tile column, instead of a separate
                           * pass over dY; < 0: none */
} commu_tn_problem;
/* number of token slices that fills the 256 CUs for this group (0: a problem is not eligible) */
int commu_gemm_tn_grouped_slices(const commu_tn_problem* probs, int nprob, int M);
/* the same for a workgroup budget: tiles x slices <= budget (the default above is 128: the launch runs on the side stream
 * beside the backward pass and leaves half of every XCD to it; the LAST launch of a pass, which the main stream ends up
 * waiting for, takes 256) */
int commu_gemm_tn_grouped_slices_budget(const commu_tn_problem* probs, int nprob, int M, int budget);
int commu_gemm_tn_bf16_grouped(const commu_tn_problem* probs, int nprob, int M, float* slabs,
                               long long slab_stride, int nslices, hipStream_t stream);
/* dst[z][r*ldd + c] = (accumulate ? dst : 0) + alpha * sum_s src[(z*nslabs + s)*stride + r*cols + c] */
int commu_reduce_slabs2d_f32(float* dst, int ldd, long long dst_batch_stride, const float* src, int rows, int cols,
                             int nslabs, size_t stride, int batch, int accumulate, float alpha, hipStream_t stream);
/* ---- MX-fp8 GEMM (BASELINE.json configs[4] "fp8 MFMA GEMMs"; the reference has no counterpart: it is the bf16 nn.Linear
 * product of commu_gemm_nt_bf16 with both operands in OCP e4m3 and one E8M0 scale per 32 k, OCP microscaling v1.0).
 * commu_quant_mxfp8: X bf16 [rows][ldx] -> Q e4m3 bytes [rows][ldq] + S scale bytes [rows][lds]; K % 32 == 0.
 *   block scale 2^(floor(log2 amax) - 8), elements rounded to nearest even, saturating at +-448.
 * commu_gemm_nt_mxfp8: C bf16 [M][ldc] = A . B^T with the epilogue flags BIAS, RELU, DROPOUT, RESID of
 *   commu_gemm_nt_bf16 (same order, same dropout element index), fp32 accumulation on
 *   v_mfma_scale_f32_16x16x128_f8f6f4; K % 128 == 0, lda/ldb % 16 == 0, ldsa/ldsb % 4 == 0, ldc % 8 == 0, else -22.
 * Tolerance against the bf16 path is set by e4m3's 3 mantissa bits (tests/test_fp8_gemm_gpu.py states it). */
int commu_quant_mxfp8(const void* X, int ldx, void* Q, int ldq, void* S, int lds, int rows, int K, hipStream_t stream);
int commu_gemm_nt_mxfp8(const void* A, int lda, const void* SA, int ldsa, const void* B, int ldb, const void* SB, int ldsb,
                        void* C, int ldc, int M, int N, int K, const float* bias, const void* resid, int ldr, int flags,
                        unsigned drop_seed, float drop_p, hipStream_t stream);

/* cropping form for zero-padded models (d_head 50 -> 64 ...): the slabs hold the padded [rg*rp, cg*cp] product, its
 * [rt, ct] blocks go to dst [rg*rt, cg*ct]:
 *   dst[((a*rt + r)*cg + c)*ct + k] (+)= alpha * sum_s src[s*stride + ((a*rp + r)*cg + c)*cp + k] */
int commu_reduce_slabs_crop_f32(float* dst, const float* src, int rg, int rt, int rp, int cg, int ct, int cp,
                                int nslabs, size_t stride, int accumulate, float alpha, hipStream_t stream);
/* all reductions of a grouped weight-gradient launch in one: item z applies the cropping form to slabs + src_off with
 * its own destination (plain reduction of [rows, cols] out of a padded [rows_p, cols] block: (1, rows, rows_p, 1, cols,
 * cols)); nitems <= 8 */
typedef struct commu_reduce_item {
    float* dst;
    long long src_off;          /* element offset of the item's block inside a slab */
    int rg, rt, rp, cg, ct, cp;
} commu_reduce_item;
int commu_reduce_slabs_group_f32(const commu_reduce_item* items, int nitems, const float* slabs, int nslabs, size_t stride,
                                 int accumulate, float alpha, hipStream_t stream);
/* dst[i] = (accumulate ? dst[i] : 0) + alpha * sum_s src[s*stride + i] */
int commu_reduce_slabs_f32(float* dst, const float* src, size_t n, int nslabs, size_t stride,
                           int accumulate, float alpha, hipStream_t stream);

/* Zero-padding contract of the row-wise kernels below (embed_fwd, posemb_fwd, layernorm_fwd/bwd): a row may be
 * wider than its D features (pitch ld > D, D % 4 == 0); columns [D, min(ld, roundup(D, 64))) are PADDING: ignored
 * on input, written as zeros on output, so the next GEMM can contract over the padded width (d_model 500 -> 512). */
/* ---- embedding (AdaptiveEmbedding.forward, model.py:409-420) and its gradient */
/* (drop_p > 0: dropout of the scaled embedding, `core_out = self.drop(word_emb)`, model.py:585;
 *  every dropout in this ABI is the counter-based mask keep(seed, element index): a 32-bit integer hash of the
 *  index, KEYED by the seed between its two multiply rounds -- see common.h drop_keep and DESIGN.md) */
/* (a token id outside [0, V) -- the reference raises IndexError -- yields a NaN row, hence a NaN loss; likewise
 *  commu_ce_fwd returns NaN for a target outside [0, V): no out-of-bounds access, no silent garbage) */
int commu_embed_fwd(const int64_t* tok, const float* E, void* out_bf16, int ldo, int ntok, int D, int V,
                    float scale, unsigned drop_seed, float drop_p, hipStream_t stream);
int commu_embed_bwd(const int64_t* tok, const void* dX_bf16, int ldx, float* dE, int ntok, int D, int V,
                    float scale, int accumulate, unsigned drop_seed, float drop_p, hipStream_t stream);
/* the same from a token order: perm = stable argsort of tok (int64 [ntok]), offs[v] = first position of id v in the sorted
 * list (int64 [V+1]; ids outside [0, V) fall outside offs[0]..offs[V] and contribute nothing).  Two passes over the
 * sorted list: every wave sums 32 consecutive sorted tokens run by run into the fp32 workspace `ws`
 * (commu_embed_bwd_ws_rows(ntok, V) rows of D floats), then dE[v] (+)= scale * (sum of v's partial rows).  The work of a
 * wave does not depend on how often an id occurs (the pad / start id is a quarter of a real batch); fixed summation order,
 * no atomics. */
int commu_embed_bwd_ws_rows(int ntok, int V);
/* (perm, offs) of the line above from the token ids themselves: a stable counting sort (V <= 1024; -22 otherwise: sort
 * with the framework).  ws: 64 * V ints of scratch.  ids outside [0, V) are left out: offs[V] = number of valid tokens,
 * perm[offs[V] ..) = 0. */
int commu_token_order(const int64_t* tok, int ntok, int V, int64_t* perm, int64_t* offs, int* ws, hipStream_t stream);
int commu_embed_bwd_sorted(const int64_t* perm, const int64_t* offs, const void* dX, int ldx, float* ws, int ntok, int D,
                           int V, float* dE, float scale, int accumulate, unsigned drop_seed, float drop_p,
                           hipStream_t stream);
/* sinusoid table by distance d: out[d] = [sin(p f) | cos(p f)], p = d, or min(d, clamp_len) when clamp_len > 0
 * (PositionalEmbedding, model.py:136-152; cfg.MODEL.clamp_len, model.py:581-582) */
int commu_posemb_fwd(const float* inv_freq, void* out_bf16, int ld, int K, int D, int clamp_len, unsigned drop_seed,
                     float drop_p, hipStream_t stream);

/* ---- LayerNorm (nn.LayerNorm at model.py:171,214; applied :179,352) */
/* y_drop (optional): a second output dropout(y) (the final `self.drop(core_out)`, model.py:601) */
int commu_layernorm_fwd(const void* z, int ldz, const float* gamma, const float* beta, void* y, int ldy,
                        float* mean, float* rstd, int rows, int D, float eps, void* y_drop, int ldyd,
                        unsigned drop_seed, float drop_p, hipStream_t stream);
/* GELU (exact erf form of torch.nn.functional.gelu) as a NON-DEFAULT FFN activation -- the reference's PositionwiseFF uses
 * ReLU (model.py:163-169), which stays the default and the parity mode; BASELINE.json's north star names a GELU-FFN:
 *   fwd: out[m,n] = dropout(gelu(z[m,n]))          bwd: dz[m,n] = dy[m,n] * keep/(1-p) * gelu'(z[m,n])
 * bf16 rows, row strides % 8 == 0 and >= cols rounded up to 8 (pad columns are written as zero); the dropout mask is the
 * GEMM epilogue's at that site: keep(drop_seed, m * cols + n). */
int commu_gelu_fwd(const void* z, int ldz, void* out, int ldo, int rows, int cols, unsigned drop_seed, float drop_p,
                   hipStream_t stream);
int commu_gelu_bwd(const void* dy, int lddy, const void* z, int ldz, void* dz, int lddz, int rows, int cols,
                   unsigned drop_seed, float drop_p, hipStream_t stream);
int commu_layernorm_bwd_nblocks(int rows);
/* part: [nblocks][3][D] partial column sums of (dy*xhat, dy, dz) */
/* dz_masked (optional): dz through the dropout that followed the Linear feeding this LayerNorm
 * (dz * keep/(1-p)); when given, part[:,2] holds ITS column sums (that Linear's bias gradient). */
int commu_layernorm_bwd(const void* dy, int lddy, const void* z, int ldz, const float* mean,
                        const float* rstd, const float* gamma, void* dz, int lddz, float* part, int rows,
                        int D, void* dz_masked, int lddzm, unsigned drop_seed, float drop_p,
                        hipStream_t stream);
/* the same with the incoming gradient dy + dy2 (both bf16, summed in fp32): the gradient of a residual branch as a second
 * addend of the LayerNorm backward instead of an auxiliary operand of the GEMM that produced dy (model.py:174-181,348-352) */
int commu_layernorm_bwd_add(const void* dy, int lddy, const void* dy2, int lddy2, const void* z, int ldz,
                            const float* mean, const float* rstd, const float* gamma, void* dz, int lddz, float* part,
                            int rows, int D, void* dz_masked, int lddzm, unsigned drop_seed, float drop_p,
                            hipStream_t stream);
/* dgamma += sum_blocks part[:,0,:], dbeta += part[:,1,:], dbias += part[:,2,:] (each destination optional) */
int commu_layernorm_bwd_reduce(const float* part, int nblk, int D, float* dgamma, float* dbeta, float* dbias,
                               hipStream_t stream);
/* out[c] += alpha * sum_r X[r,c]   (bias gradients).  No atomics, fixed summation order: row slabs are summed into the fp32
 * workspace `ws` (commu_colsum_slabs(rows, cols, element bytes) rows of round_up(cols, 8) floats; 0 rows: not needed),
 * then over the slabs.  Rows must be 16-byte aligned chunks (ldx % 8 == 0 for bf16, % 4 for fp32, round_up(cols) <= ldx);
 * -22 otherwise or when the workspace is missing. */
int commu_colsum_slabs(int rows, int cols, int elem_bytes);
int commu_colsum_bf16(const void* X, int ldx, int rows, int cols, float* out, float* ws, int ws_rows, float alpha,
                      hipStream_t stream);
int commu_colsum_f32(const float* X, int ldx, int rows, int cols, float* out, float* ws, int ws_rows, float alpha,
                     hipStream_t stream);
/* The final passes of a whole backward step in one launch.  A task adds into out[0 .. cols) the column sums of its sources
 * src[src_begin .. src_end) -- fp32 matrices of `rows` rows with row stride ldx (partial rows of commu_colsum_slab_pass or
 * of commu_layernorm_bwd, per-tile sums of the attention kernels) -- each times its alpha, in the order given.  Two tasks
 * must not share an output.  At most COMMU_COLSUM_MAX_TASKS tasks / COMMU_COLSUM_MAX_SOURCES sources per call. */
#define COMMU_COLSUM_MAX_TASKS 48
#define COMMU_COLSUM_MAX_SOURCES 64
typedef struct commu_colsum_source {
    const float* X;
    int ldx, rows;
    float alpha;
    int pad_;
} commu_colsum_source;
typedef struct commu_colsum_task {
    float* out;
    int cols, src_begin, src_end, pad_;
} commu_colsum_task;
int commu_colsum_group_f32(const commu_colsum_task* tasks, int ntask, const commu_colsum_source* srcs, int nsrc,
                           hipStream_t stream);
/* the slab pass of commu_colsum_bf16 / _f32 on its own (elem_bytes 2 / 4): returns the number of partial rows written to
 * ws (row stride = cols rounded up to 8 / 4), 0 when the input is small fp32 and is summed directly, < 0 on error */
int commu_colsum_slab_pass(const void* X, int elem_bytes, int ldx, int rows, int cols, float* ws, int ws_rows,
                           hipStream_t stream);

/* ---- output layer loss (ProjectedAdaptiveLogSoftmax.forward, n_clusters == 0: model.py:64-73) */
int commu_ce_fwd(const float* logits, int ldl, const int64_t* target, float* nll, float* lse, int rows,
                 int V, hipStream_t stream);
int commu_ce_bwd(const float* logits, int ldl, const int64_t* target, const float* lse, const float* g,
                 void* dlogits_bf16, int ldd, int rows, int V, hipStream_t stream);
/* out[0] = scale * mean(nll[target != pad])   (train.py:148-149); keeps the count in cnt_ws; NaN when every
 * target is the pad id (mean of an empty selection, as in the reference) */
int commu_masked_mean(const float* nll, const int64_t* target, int n, int pad, float scale, float* sum_ws,
                      int* cnt_ws, float* out, hipStream_t stream);
int commu_loss_grad(const int64_t* target, int n, int pad, const int* cnt_ws, float scale, float* g,
                    hipStream_t stream);
/* Column-grouped forms for a [T][B] time-major loss whose B columns are B / Bc micro-batches of Bc columns (the reference's
 * `batch_chunk`, train.py:136-155): out[0] = scale * sum_g mean(nll[target != pad] in group g) (scale = 1 / batch_chunk),
 * g[m] = scale * (target[m] != pad) / count_{group of column m % B}.  One forward / backward over all columns then equals
 * the reference's loop over micro-batches.  sum_ws / cnt_ws: B / Bc (<= 16) words; sum_all (optional): sum over all groups. */
int commu_masked_mean_groups(const float* nll, const int64_t* target, int n, int pad, float scale, int B, int Bc,
                             float* sum_ws, int* cnt_ws, float* out, float* sum_all, hipStream_t stream);
int commu_loss_grad_groups(const int64_t* target, int n, int pad, const int* cnt_ws, float scale, int B, int Bc, float* g,
                           hipStream_t stream);

/* ---- optimiser (clip_grad_norm_ + optim.Adam, train.py:159-165,442) on flat fp32 buffers
 * (commu_adam_step / _dev: the 16-byte vector body needs p, g, m, v 16-byte aligned and p_bf16 8-byte aligned or null; any other
 *  slice runs the element-wise body of the same kernel -- same results) */
int commu_grad_norm(const float* g, size_t n, float* part, int npart, float* out_norm, hipStream_t stream);
int commu_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, size_t n, float lr,
                    float beta1, float beta2, float eps, int step, const float* gnorm, float clip,
                    hipStream_t stream);
/* The same step with {lr, 1 - beta1^t, 1 - beta2^t} read from DEVICE memory (`scal`, three floats): the form a
 * hipGraph-captured training step replays; commu_adam_bias_corrections (host-only, no stream) gives the two
 * corrections exactly as commu_adam_step computes them. */
int commu_adam_step_dev(float* p, const float* g, float* m, float* v, void* p_bf16, size_t n, const float* scal,
                        float beta1, float beta2, float eps, const float* gnorm, float clip, hipStream_t stream);
int commu_adam_bias_corrections(float beta1, float beta2, int step, float* out2);
/* Dropout seed salt: every dropout site of every kernel uses (its seed argument + salt); the salt is 0 until this call
 * loads it from device memory (src == NULL: back to 0).  Kernel arguments of a captured hipGraph are frozen, the salt
 * is not: a captured training step starts with this call and draws fresh masks on every replay. */
int commu_set_seed_salt(const unsigned* src, hipStream_t stream);
/* g *= min(1, clip / (gnorm[0] + 1e-6)) */
int commu_scale_clip_f32(float* g, size_t n, const float* gnorm, float clip, hipStream_t stream);
int commu_cast_f32_bf16(const float* in, void* out, size_t n, hipStream_t stream);
int commu_cast_bf16_f32(const void* in, float* out, size_t n, hipStream_t stream);
int commu_transpose_bf16(const void* in, int ldi, void* out, int ldo, int rows, int cols, hipStream_t stream);
int commu_transpose_f32_bf16(const float* in, int ldi, void* out, int ldo, int rows, int cols,
                             hipStream_t stream);
int commu_copy_bf16(const void* src, void* dst, size_t n, hipStream_t stream);
/* Up to 32 bf16 transposes in ONE launch: out_i[c][r] = in_i[r][c] (the W^T shadows of every Linear weight that the
 * dX = dY . W GEMMs read, autograd of model.py:205,212,164,167,46: 26 per optimiser step at 6 layers, each a launch of a
 * few dozen workgroups before).  -22 for n > 32. */
typedef struct commu_transpose_item {
    const void* in;
    void* out;
    int ldi, ldo, rows, cols;
} commu_transpose_item;
int commu_transpose_group_bf16(const commu_transpose_item* items, int n, hipStream_t stream);

/* K9  memory update (reference commu/model/model.py:507-538 _update_mems): for each of `layers` layers
 *   out[l] = [ mems[l][mem_skip : mem_skip+keep) ; hids[l][hid_skip : hid_skip+take) ]
 * all counts/strides in bf16 ELEMENTS and multiples of 8 (a time step is a whole [B, Dp] slab); one launch.
 * -22 on a misaligned count or a missing pointer. */
int commu_mems_update(const void* hids, size_t hid_stride, size_t hid_skip, size_t take, const void* mems,
                      size_t mem_stride, size_t mem_skip, size_t keep, void* out, size_t out_stride, int layers,
                      hipStream_t stream);

/* ---- relative-position attention with XL memory
 * (RelPartialLearnableMultiHeadAttn.forward, model.py:313-345; _rel_shift :251-259; mask :549-574) */
typedef struct commu_attn_desc {
    const void* q;   /* bf16, row i: q + (i*B + b)*ld_qkv + h*DH, i in [0,T)   (current segment) */
    const void* k;   /* bf16, row j: k + (j*B + b)*ld_qkv + h*DH, j in [0,T+M) (memory first)    */
    const void* v;
    const void* rd;  /* bf16 [T+M][ld_rd]: r_net(sinusoid(distance d)), distance d = i + M - j   */
    const float* r_w_bias;          /* [H][DH]  (u) */
    const float* r_r_bias;          /* [H][DH]  (v) */
    const unsigned char* reset;     /* [B] reset_mems flags or NULL (model.py:574) */
    int ld_qkv, ld_rd, ld_o;
    int T, M, B, H, DH;
    int same_length, sshift;        /* same_length mask: j <= i - sshift is masked (model.py:549-568) */
    float scale;                    /* 1/sqrt(d_head), model.py:216 */
    float drop_p;                   /* attention-probability dropout (self.dropatt, model.py:337); 0 = off */
    unsigned drop_seed;
} commu_attn_desc;

/* out: bf16 [T*B][ld_o]; lse: fp32 [B][H][T] (natural log).  qu2/qv2 (both or neither): bf16
 * [T*B][H*DH] copies of (q + r_w_bias) and (q + r_r_bias) times scale*log2(e), the operands the
 * backward kernels re-use. */
int commu_relattn_fwd(const commu_attn_desc* d, void* out, float* lse, void* qu2, void* qv2,
                      hipStream_t stream);
/* The same forward pass, additionally SAVING its probabilities for the backward pass (d_head 64, generation-3 forward
 * only; -22 otherwise): pf = commu_attn_pf_bytes(T, M, B, H) bytes, uninitialised -- per (batch, head, 32-query block,
 * 32-key sub-tile the forward visits) a 2176-byte tile: exp2(s - m) as bf16 in the forward kernel's accumulator order,
 * m = the query's running maximum at that sub-tile, the dropout decision in the sign bit, then the 32 values of m.
 * commu_attn_bwd_desc.pf hands it to the query-stationary backward kernel, which then neither recomputes
 * (q + u) . k, the band product / rel-shift, the masks, exp nor the dropout hash (autograd of model.py:313-345). */
int commu_relattn_fwd_save(const commu_attn_desc* d, void* out, float* lse, void* qu2, void* qv2, void* pf,
                           hipStream_t stream);
long long commu_attn_pf_bytes(int T, int M, int B, int H);
/* Forward kernel generation for d_head 64 (process-wide; returns the previous value).  0 (default) and 3: the 32x32-MFMA /
 * transposed-score kernel (relattn3.hip), with or without attention dropout; 2: the 16x16-layout kernel.  Every attention
 * kernel -- both forward generations and the backward family -- regenerates the same dropout mask (relattn.hip DropLane: one
 * mixed word per 2x2 cell of a 32x32 block, one multiply-add per element), so any forward pairs with the backward. */
int commu_attn_fwd_generation(int gen);
/* Backward kernel pair for d_head 64 with a P scratch (process-wide; returns the previous value).  3: relattn_kv3.hip
 * (32 keys per wave on the 32x32 MFMA) behind the 16x16 query-stationary kernel; 4: both kernels on the 32x32 MFMA
 * (relattn_q3.hip: 32 query rows per wave, transposed scores; relattn_kv3.hip transposes its P blocks through LDS);
 * 2: the 16x16-layout pair; 0: the build's default.  The setting fixes the block order in
 * which commu_relattn_bwd_q writes the scratch: both launches of a backward pass must see the same value. */
int commu_attn_bwd_kv_generation(int gen);

/* Backward of commu_relattn_fwd (autograd of model.py:313-345).  Produces dk, dv, the AC part of dq
 * and dS indexed by distance; the caller finishes with two GEMMs per head:
 *   dq[:, h] = dq_ac[:, h] + dsk[h] . Rd[:, h]          (commu_gemm_nt_bf16, RESID)
 *   dRd[:, h] = dsk[h]^T . qv2[:, h] / (scale*log2 e)   (commu_gemm_tn_bf16)
 * and d r_w_bias = colsum(dq_ac), d r_r_bias = colsum(dq - dq_ac). */
typedef struct commu_attn_bwd_desc {
    const void* dout;     /* gradient of the forward output, bf16 [T*B][ld_o] */
    const float* lse;     /* [B][H][T] from the forward */
    const float* delta;   /* [B][H][T]  sum_f dO*O  (commu_attn_delta) */
    const void* qu2;      /* from the forward */
    const void* qv2;
    void* dq_ac;          /* bf16 [T*B][H*DH] */
    void* dk;             /* bf16 rows like k with ld_dqkv */
    void* dv;
    void* dsk;            /* bf16 [H][T*B][ld_dsk]: dS by distance d = i + M - j.  dsk_wedge == 0: ZERO-INITIALISED by
                             the caller; dsk_wedge > 0: uninitialised -- the kernel writes columns 0..i+M of row
                             (i, b) and zeros columns i+M+1 .. i+M+dsk_wedge; the rest is never read by the band
                             GEMMs (only valid without same_length / reset masks) */
    float* du_part;       /* [B*du_rows][H*DH] column sums of dq_ac per query tile */
    int ld_dqkv, ld_dsk;
    int du_rows;          /* = ceil(T / commu_attn_bwd_qrows(T)) */
    int dsk_wedge;
    int dsk_tiled;        /* != 0: dsk is [H][T*B/64][ld_dsk/128] tiles of [64 rows][128 distances] (the layout
                             commu_relattn_bwd_band reads; needs (T*B) % 64 == 0, ld_dsk % 128 == 0); 0: row-major */
    void* p_scratch;      /* NULL, or (d_head 64 only) commu_attn_p_scratch_elems(T, M, B, H) bf16 elements, uninitialised:
                             the query-stationary kernel stores the probabilities it recomputes there and the
                             key-stationary kernel, which must then run AFTER it on the same stream, reads them
                             back instead of recomputing (q+u).k, the band product / rel-shift, masks and exp.  The
                             buffer is private to that pair of launches: its block order is the key-stationary kernel's
                             read order, and with attention dropout a stored value is NEGATIVE where the mask drops the
                             element (the keep decision travels in the sign; that kernel does not hash) */
    const void* o;        /* NULL, or the forward output (bf16 [T*B][ld_o]): the query-stationary kernel then computes
                             delta[b,h,i] = sum_d o . dout itself (commu_attn_delta is not needed) and WRITES it to
                             `delta` for the key-stationary kernel, which must run after it on the same stream */
    const void* pf;       /* NULL, or the buffer commu_relattn_fwd_save filled for this call (needs p_scratch, d_head 64) */
} commu_attn_bwd_desc;
long long commu_attn_p_scratch_elems(int T, int M, int B, int H);
int commu_attn_bwd_qrows(int T);
int commu_relattn_bwd(const commu_attn_desc* d, const commu_attn_bwd_desc* e, hipStream_t stream);
/* the two kernels of commu_relattn_bwd on their own: query-stationary (dq_ac, dsk, du_part) and key-stationary
 * (dk, dv); same descriptors */
int commu_relattn_bwd_q(const commu_attn_desc* d, const commu_attn_bwd_desc* e, hipStream_t stream);
int commu_relattn_bwd_kv(const commu_attn_desc* d, const commu_attn_bwd_desc* e, hipStream_t stream);
/* The two per-head GEMMs above fused into ONE pass over dsk (band.hip): dq = dq_ac + dsk . Rd (bf16, written in
 * place with row stride ld_dq) and slabs[(h * P + p)][Kp][64] = partial dsk^T . qv2 of token-slice pair p, with
 * P = commu_attn_band_pairs(T, B, H) and Kp = T + M rounded up to 512 (the kernel runs one pass per 512 distances,
 * keeping that part of Rd resident in LDS); the caller finishes with
 *   commu_reduce_slabs2d_f32(dRd, ld, 64, slabs, K, 64, P, Kp * 64, H, 0, 1 / (scale * log2 e)).
 * dsk must be TILED (commu_attn_bwd_desc.dsk_tiled).  Needs DH == 64, T + M <= 4096, ld_dsk % 128 == 0, (T * B) % 64 == 0
 * and >= 8192 (else -22: use the two GEMMs on a row-major dsk).
 * band != 0: rows are causal (see commu_gemm_nt_bf16_batched, tri_B = B, tri_M = M) and only the chunks of 256
 * distances that reach the causal edge are read; what lies beyond the edge inside them must be zero.
 * commu_attn_band_pairs: token-slice pairs per head = min(32, 256 / H) -- H * P workgroups fill the 256 CUs once; with 16
 * heads, 16 pairs write half the slab bytes of 32 -- 0: shape not taken.  commu_attn_band_slabs(T, B) = its upper bound 32
 * (or 0), for callers that size a buffer before the head count is known. */
int commu_attn_band_slabs(int T, int B);
int commu_attn_band_pairs(int T, int B, int H);
int commu_relattn_bwd_band(const void* dsk, int ld_dsk, const void* rd, int ld_rd, const void* qv2, int ld_qv,
                           const void* dq_ac, int ld_ac, void* dq, int ld_dq, float* slabs, int T, int M, int B,
                           int H, int DH, int band, hipStream_t stream);
int commu_attn_delta(const void* o, const void* dout, int ld, float* delta, int T, int B, int H, int DH,
                     hipStream_t stream);
/* dst[((b*H+h)*DH+f)*W + off + j] = src[(j*B+b)*ld + h*DH + f] (+ bias[h*DH+f]); zero elsewhere */
int commu_transpose_heads(const void* src, int ld, const float* bias, void* dst, int J, int B, int H,
                          int DH, int W, int off, hipStream_t stream);

/* ---- sampling step of the decode loop (InferenceTask.calc_probs / apply_sampling / infer_token,
 * commu/midi_generator/midi_inferrer.py:209-237), one wave per sequence:
 *   logits[b][1:V] /= temperature IN PLACE (quirk Q5), softmax, pad column 0 -> 0 (Q6), keep the
 *   top_k, zero wrong[b][id] != 0, renormalise, draw by inverse CDF with uniforms[b].
 *   temperature == 0: one-hot argmax.  token[b] = -1 when nothing can be drawn (Q12).
 *   active (optional): only sequences with active[b] != 0 are processed.
 *   probs_out (optional): the [nseq][ldp] distribution that was drawn from. */
int commu_sample_topk(float* logits, int ld, int nseq, int V, const unsigned char* wrong, int ldw,
                      const float* uniforms, const unsigned char* active, float temperature, int top_k,
                      int* token, float* probs_out, int ldp, hipStream_t stream);
/* the same with a nucleus ("top-p") filter after the top-k / rejected-token step -- an extra mode, the reference has none
 * (generate.py:43-44 offers top_k and temperature only): in order of decreasing probability (ties: lowest id first) a
 * token is kept while the probability mass before it is < top_p; survivors renormalised.  top_p >= 1: off. */
int commu_sample_topk_topp(float* logits, int ld, int nseq, int V, const unsigned char* wrong, int ldw,
                           const float* uniforms, const unsigned char* active, float temperature, int top_k,
                           float top_p, int* token, float* probs_out, int ldp, hipStream_t stream);

/* ---- single-token decode step with a K/V cache (forward_generate with qlen 1, model.py:606-628, as
 * called by midi_inferrer.py:199-207).  Caches are bf16 [B][H][Lmax][DH] per layer; klen[b] = valid rows
 * of sequence b (ragged); the new token sits at row klen[b] and attends rows 0..klen[b]. */
int commu_decode_kv_append(const void* qkv, int ld_qkv, void* kcache, void* vcache, const int* klen,
                           const unsigned char* active, int B, int Lmax, int H, int HD, hipStream_t stream);
/* out[b, h*DH:(h+1)*DH] = softmax_j(((q+u).k_j + (q+v).Rd[klen[b]-j]) * scale) . v_j   (rd: [>= Lmax][ld_rd]) */
/* append != 0: the kernel first writes the new token's K and V (from qkv) to cache row klen[b] itself
 * (commu_decode_kv_append fused in; the caches are then written through kcache / vcache) */
int commu_decode_attn(const void* qkv, int ld_qkv, void* kcache, void* vcache, const void* rd,
                      int ld_rd, const float* r_w_bias, const float* r_r_bias, const int* klen,
                      const unsigned char* active, void* out, int ld_o, int B, int H, int DH, int Lmax,
                      float scale, int append, hipStream_t stream);
/* The same with the keys of a (sequence, head) pair split over up to nsplit (<= 16) workgroups of >= 512 keys each -- for
 * long memories with few live sequences, where one workgroup per pair streams its ~1 MB alone.  split_ws: B * H * nsplit *
 * (DH + 2) floats of scratch; split_cnt: B * H words, ZERO before the first call (the kernel leaves them zero).  Pairs
 * with at most 512 keys take the unsplit path inside the same launch. */
int commu_decode_attn_split(const void* qkv, int ld_qkv, void* kcache, void* vcache, const void* rd, int ld_rd,
                            const float* r_w_bias, const float* r_r_bias, const int* klen, const unsigned char* active,
                            void* out, int ld_o, int B, int H, int DH, int Lmax, float scale, int append, int nsplit,
                            float* split_ws, unsigned* split_cnt, hipStream_t stream);
/* klen[b] += advance[b]  (a step whose memory the reference discards does not advance: quirk Q3) */
int commu_decode_advance(int* klen, const unsigned char* advance, int B, int Lmax, hipStream_t stream);
/* Everything of a decode-step layer that follows its attention, as ONE launch (csrc/decode_tail.hip):
 *   z1 = vec . Wo^T + h;  a = LN1(z1);  hid = relu(a . W1^T + b1);  z2 = hid . W2^T + b2 + a;  h_out = LN2(z2)
 *   (o_net + residual + LayerNorm model.py:344-352, PositionwiseFF model.py:163-179)
 * and then out_n = h_out . Wn^T: the NEXT layer's qkv_net (logits == 0: Nn = 3 HD, bf16, model.py:297-299) or the
 * tied-embedding logits (logits != 0: + bn, fp32, columns < Nn, model.py:64-73; rows with active[row] == 0 are not
 * written -- active may be null).  vec [B][HD], h / h_out [B][D] bf16;
 * z1 / z2 [B][D] and hid [B][DI] are dense bf16 hand-off buffers that belong to THIS layer; sync:
 * commu_decode_tail_sync_words() arrival counters that must be ZERO on entry (one set per launch of a step); *err
 * becomes non-zero when a workgroup gave up waiting (the results of that launch are then invalid).
 * commu_decode_tail_supported(B, D, DI, HD) names the shapes this build takes -- (512, 1024, 8 or 10 heads x 64) and the
 * wide (1024, 2048, 16 x 64), up to 64 sequences -- (others: -22; callers use the per-Linear launches).  d_ln: LayerNorm width (<= D; zero-padded models). */
int commu_decode_tail_supported(int B, int D, int DI, int HD);
/* diagnostics: following layer-tail / head launches write 100 MHz timestamps of their phase boundaries to
 * buf[workgroup][16] (device memory, 128 x 16 words; null: off) */
int commu_decode_tail_trace(unsigned long long* buf);
int commu_decode_tail_sync_words(void);
int commu_decode_layer_tail(const void* vec, int ld_vec, const void* h, int ld_h, const void* Wo_packed,
                            const void* W1_packed, const float* b1, const void* W2_packed, const float* b2,
                            const float* g1, const float* be1, float eps1, const float* g2, const float* be2, float eps2,
                            int d_ln, const void* Wn_packed, int Nn, const float* bn, int logits,
                            const unsigned char* active, void* z1, void* hid, void* z2, void* h_out, int ld_ho,
                            void* out_n, int ld_on, int B, int D, int DI, int HD, unsigned* sync, unsigned* err,
                            hipStream_t stream);
/* The four weights of a layer tail (and the head's) are read from PACKED copies: the fragments of one workgroup and
 * wave are consecutive, so every wave instruction reads 1 KB of consecutive bytes.  commu_decode_tail_pack writes the
 * packed copy of a bf16 [N][K] weight (row stride ldw, K % 128 == 0; rows >= N of the last tiles are zero) into `out`
 * (commu_decode_tail_pack_bytes(N, K) bytes); it must be repeated whenever the weight changes. */
long long commu_decode_tail_pack_bytes(int N, int K);
int commu_decode_tail_pack(const void* W, int ldw, int N, int K, void* out, hipStream_t stream);
/* First launch of a decode step on the same workgroup layout: h_out = E[tok] * scale (word embedding, model.py:409-420;
 * fp32 table [V][d_true], ids outside [0, V) give NaN rows) and qkv = h_out . Wqkv^T (layer 0's qkv_net); also clears
 * zero_words[0 .. n_zero) -- the arrival counters of the step's commu_decode_layer_tail launches. */
int commu_decode_head(const int64_t* tok, const float* E, int d_true, int V, float scale, const void* Wqkv_packed,
                      void* h_out, int ld_ho, void* qkv, int ld_qkv, int B, int D, int DI, int HD,
                      unsigned* zero_words, int n_zero, hipStream_t stream);

/* ---- device-resident chord / bar forcing of the decode loop (InferenceTask.generate_sequence +
 * TeacherForceTask, commu/midi_generator/midi_inferrer.py:239-320, :16-144): per-sequence state records of
 * commu_forcing_state_ints() int32 each, fields in this order:
 *   len, forced (-1: none), redo, first, filled, done, failed, iters, nbar, nchord, cur, length_fit, ndraw, ntrace.
 * seq: int32 [B][ld_seq] token buffers; chord_tok / chord_pos: int32 [B][ld_chord] (the progression to force);
 * wrong: uint8 [B][729] rejected-chord bitmap; utable: fp32 [B][ld_u] uniform variates, one per draw.
 * commu_forcing_pre decides this iteration's model step and draw: tok (int64 [B]: token fed), active (step?),
 * keep (does the step's memory stay? quirk Q3), draw (is a token drawn?), uni (its variate); trace (optional,
 * int32 [B][ld_trace]) records (token, keep) of every model step.  commu_forcing_post applies the drawn token
 * (token[b] < 0: nothing could be drawn, Q12).  live (optional): += number of unfinished sequences. */
int commu_forcing_state_ints(void);
int commu_forcing_pre(int* state, int* seq, int ld_seq, const int* chord_tok, const int* chord_pos, int ld_chord,
                      unsigned char* wrong, const float* utable, int ld_u, int max_iters, long long* tok,
                      unsigned char* active, unsigned char* keep, unsigned char* draw, float* uni, int* trace,
                      int ld_trace, int B, hipStream_t stream);
/* klen / keep (optional): the cache lengths of the decode step advance here (klen[b] += keep[b], capped at lmax - 1),
 * which saves the separate commu_decode_advance launch */
int commu_forcing_post(int* state, int* seq, int ld_seq, const int* chord_pos, int ld_chord, unsigned char* wrong,
                       const unsigned char* draw, const int* token, int* live, int* klen, const unsigned char* keep,
                       int lmax, int B, hipStream_t stream);
/* The three per-sequence stages that follow the model step as ONE launch, in this order and with the meaning of the
 * separate entry points: commu_sample_topk_topp (active = draw, wrong [B][729]) -> commu_forcing_post (live = null) ->
 * commu_forcing_pre (the decision of the NEXT iteration). */
/* diagnostics: following commu_decode_sample_post_pre launches write 100 MHz timestamps buf[sequence][4] = kernel start,
 * after the sampling step, after post, after pre (device memory; null: off) */
int commu_decode_loop_trace(unsigned long long* buf);
int commu_decode_sample_post_pre(float* logits, int ld, int V, unsigned char* wrong, float temperature, int top_k,
                                 float top_p, int* token, float* probs_out, int ldp, int* state, int* seq, int ld_seq,
                                 const int* chord_tok, const int* chord_pos, int ld_chord, const float* utable, int ld_u,
                                 int max_iters, long long* tok, unsigned char* active, unsigned char* keep,
                                 unsigned char* draw, float* uni, int* trace, int ld_trace, int* klen, int lmax, int B,
                                 hipStream_t stream);
/* dst[b][0:n] = src[b][0:n] where mask[b] != 0 */
int commu_copy_rows_masked_f32(float* dst, int ldd, const float* src, int lds, const unsigned char* mask, int rows,
                               int n, hipStream_t stream);

/* ---- fp32 PARITY MODE of the generation path (csrc/parity_f32.hip; model.parity_fp32 / generate.py --parity).
 * The reference computes in fp32 throughout (train.py:48 `amp = None`; no autocast in commu/model/model.py); these entry
 * points restate the forward pass on fp32 operands end to end -- fp32 master weights, activations and K/V cache, fp32
 * MFMA products, accurate expf / sinf / cosf -- so that greedy decoding (midi_inferrer.py:199-237, temperature 0) picks
 * the reference's tokens wherever the top-1 / top-2 logit gap exceeds fp32 summation-order noise (~1e-6 of the range). */
/* nn.Linear (model.py:46,164,167,205,212,278): C[M,N] = A[M,K] . B[N,K]^T (+ bias[n]) (ReLU if relu) (+ resid[m,n]); all
 * fp32 row-major; any M, N, K. */
int commu_gemm_nt_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                      const float* bias, const float* resid, int ldr, int relu, hipStream_t stream);
/* model.py:409-420: out[row][0:D] = E[tok[row]][0:D] * scale (scale = sqrt(d_model)) */
int commu_embed_f32(const long long* tok, const float* E, float* out, int ld, int rows, int D, float scale,
                    hipStream_t stream);
/* model.py:142-147 indexed by DISTANCE: out[d] = [sin(p * inv_freq) | cos(p * inv_freq)], d = 0 .. n-1, p = d or
 * min(d, clamp_len) when clamp_len > 0 (model.py:581-582) */
int commu_posemb_f32(const float* inv_freq, float* out, int ld, int n, int D, int clamp_len, hipStream_t stream);
/* nn.LayerNorm (model.py:179,352), D <= 1024 */
int commu_layernorm_f32(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, int rows,
                        int D, float eps, hipStream_t stream);
/* model.py:283-345 (RelPartialLearnableMultiHeadAttn.forward without the projections), eval mode: q [T*B rows][ld_q]
 * (row i*B+b, head h at column h*DH), element (key j, sequence b, head h, d) of k / v at base + j*stride_key +
 * b*stride_seq + h*DH + d, rd [distance][ld_rd], out [T*B][ld_o].  klen (optional, int32 [B]): memory length of each
 * sequence (ragged decode batch; otherwise M for all); reset (optional, uint8 [B]): hide the memory keys of that
 * sequence; masks of model.py:549-574 with the model's mem_len.  DH <= 64. */
int commu_relattn_f32(const float* q, int ld_q, const float* k, const float* v, long long stride_key,
                      long long stride_seq, const float* rd, int ld_rd, const float* r_w_bias, const float* r_r_bias,
                      const int* klen, const unsigned char* reset, float* out, int ld_o, int T, int M, int B, int H,
                      int DH, int same_length, int mem_len, float scale, hipStream_t stream);
/* decode step: columns [HD, 2HD) / [2HD, 3HD) of row b of the new token's projection into row klen[b] of the fp32 caches
 * kc / vc [B][Lmax][HD], for the sequences with active[b] != 0 (null: all) */
int commu_decode_kv_append_f32(const float* qkv, int ld, float* kc, float* vc, const int* klen,
                               const unsigned char* active, int B, int HD, int Lmax, hipStream_t stream);

/* library identification */
const char* commu_hip_version(void);

/* Host-side (no GPU): assemble one [T][B] int64 training batch from the flat corpus.  Column c streams sequence
 * seq[c] (-1: none) from position pos[c]: data[r][c] = tokens[offsets[seq[c]] + pos[c] + r] for r < cnt[c], target the
 * same shifted by one, `pad` elsewhere (commu/model/dataset.py:139-170).  Returns the number of non-pad targets
 * (sum of cnt), -22 for B > 1024.  Thread-safe; called without the GIL by the prefetch thread. */
long long commu_pack_batch(const int64_t* tokens, const int64_t* offsets, const int64_t* seq, const int64_t* pos,
                           const int64_t* cnt, int B, int T, int64_t pad, int64_t* data, int64_t* target);

#ifdef __cplusplus
}
#endif
#endif /* COMMU_HIP_H */
