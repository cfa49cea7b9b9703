"""Tensor-level wrappers over the C ABI (include/commu_hip.h).

PyTorch is used here only for device memory and the current HIP stream; all arithmetic is done
by the kernels in libcommu_hip.so.  Every wrapper refuses CPU tensors: there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib
from ._lib import AttnBwdDesc, AttnDesc, CommuHipError, call

EPI_BIAS, EPI_RELU, EPI_RESID, EPI_RELUMASK, EPI_OUT_F32, EPI_DROPOUT, EPI_SIGNBITS_OUT, EPI_RELUBITS = 1, 2, 4, 8, 16, 32, 64, 128


def site_seed(base_seed: int, site: int) -> int:
    """32-bit seed of one dropout site of one forward call (host-side integer hash)."""
    x = (base_seed * 0x9E3779B9 + site * 0x85EBCA6B + 0x165667B1) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def _mix32(x):
    x = x & 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x7feb352d) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 0x846ca68b) & 0xFFFFFFFF
    return x ^ (x >> 16)


def _mix32k(x, key):
    x = x & 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x7feb352d) & 0xFFFFFFFF
    x = x ^ key                               # the key enters between the two multiply rounds (common.h mix32k)
    x = x ^ (x >> 15)
    x = (x * 0x846ca68b) & 0xFFFFFFFF
    return x ^ (x >> 16)


def dropout_keep_mask(seed: int, n: int, p: float, device="cpu"):
    """The kernels' keep mask for element indices 0..n-1 (reference implementation for tests; common.h drop_word): one
    hash word per two consecutive indices -- two 24-bit multiply-add rounds around a xor-shift, keyed by mix32(seed) --, the
    even index takes the low 16 bits, the odd one the high 16, against round(p * 65536).  Kept values are scaled by
    1 / (1 - thr / 65536) in the kernels (within 8e-6 of 1 / (1 - p))."""
    M32, M24 = 0xFFFFFFFF, 0xFFFFFF
    key = int(_mix32(torch.tensor(int(seed) & M32, dtype=torch.int64)))
    k2 = (key * 0x85EBCA6B + 0x6A09E667) & M32
    idx = torch.arange(n, dtype=torch.int64, device=device)
    q = idx >> 1
    y = ((q & M24) * 0xD2B74B + key) & M32
    y = (((q >> 24) & M24) * 0xB5297A + y) & M32
    y = y ^ (y >> 13)
    w = ((y & M24) * 0x9E3779 + k2) & M32
    w = w ^ (w >> 15)
    half = torch.where((idx & 1).bool(), w >> 16, w & 0xFFFF)
    thr = min(65535, max(1, int(p * 65536.0 + 0.5))) if p > 0 else 0
    return half >= thr


BF16 = torch.bfloat16
F32 = torch.float32

# use LDS transpose reads in the dW GEMM (mode 1) unless told otherwise
TN_MODE = 1


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise CommuHipError("commu_amd kernels need GPU tensors (no CPU fallback)")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _s():
    # (the raw handle of torch's current stream: ~0.3 us, against ~8 us for torch.cuda.current_stream().cuda_stream --
    #  a training step makes ~250 kernel calls)
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def attn_dropout_keep_mask(seed: int, B: int, H: int, T: int, K: int, p: float, version: int = 3):
    """The attention kernels' keep mask [B,H,T,K] (reference implementation for tests; relattn.hip DropLane).  One mask for
    every kernel: each 32x32 block (i>>5, j>>5) of a (batch, head) has a 32-bit key k1 from the strong hash; inside the block
    one mixed word per 2x2 cell, y = ((i&31)>>1 << 4 | (j&31)>>1) * C1 + k1, y ^= y >> 12, y &= 0xFFFFFF, and one
    multiply-add per element with the constants of its place in the cell, w = y * CM[i&1][j&1] + k1 * KA[i&1][j&1] +
    KB[i&1][j&1]; keep = w >= round(p * 65536) << 16.  (`version` is accepted for older call sites and ignored.)"""
    thr = max(1, int(p * 65536.0 + 0.5)) if p > 0 else 0
    out = torch.empty(B, H, T, K, dtype=torch.bool)
    rows = torch.arange(T, dtype=torch.int64)[:, None]
    cols = torch.arange(K, dtype=torch.int64)[None, :]
    M32 = 0xFFFFFFFF
    CM = ((0x9E3779, 0x85EBCB), (0xC2B2AF, 0xB5297B))
    KA = ((0x85EBCA6B, 0xC2B2AE35), (0x27D4EB2F, 0x165667B1))
    KB = ((0x6A09E667, 0xBB67AE85), (0x3C6EF372, 0xA54FF53A))
    blk = ((rows >> 5) << 16) | (cols >> 5)
    xc = (((((rows & 31) >> 1) << 4) | ((cols & 31) >> 1)) * 0xD2B74B) & M32
    ri = (rows & 1).expand(T, K)
    ci = (cols & 1).expand(T, K)
    for b in range(B):
        for h in range(H):
            key_bh = int(_mix32(torch.tensor((seed + (b * H + h) * 0x9E3779B1) & M32, dtype=torch.int64)))
            k1 = _mix32k(blk, key_bh)
            y = (xc + k1) & M32
            y = (y ^ (y >> 12)) & 0xFFFFFF
            w = torch.zeros(T, K, dtype=torch.int64)
            for r_ in range(2):
                for c_ in range(2):
                    sel = (ri == r_) & (ci == c_)
                    wv = (y * CM[r_][c_] + ((k1 * KA[r_][c_] + KB[r_][c_]) & M32)) & M32
                    w = torch.where(sel, wv, w)
            out[b, h] = w >= (thr << 16)
    return out, 1.0 - thr / 65536.0


def attn_fwd_generation(gen: int) -> int:
    """commu_attn_fwd_generation: 0 automatic, 2 / 3 force a forward kernel family for d_head 64; returns the previous value."""
    return call("commu_attn_fwd_generation", int(gen))


def attn_bwd_kv_generation(gen: int) -> int:
    """commu_attn_bwd_kv_generation: 0 automatic; 2 / 3 force a key-stationary backward kernel for d_head 64 with stored
    probabilities; 4: both backward kernels on the 32x32 MFMA (relattn_q3.hip + relattn_kv3.hip); returns the previous value."""
    return call("commu_attn_bwd_kv_generation", int(gen))


def _rowmajor2d(t: torch.Tensor, name: str):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a 2-D tensor with unit column stride, got {tuple(t.shape)} / {t.stride()}")
    return t.stride(0)


def gemm_nt_layers(A, Bflat, N, K, stride, batch, out=None):
    """out[z] = A @ B_z^T for z < batch, B_z = Bflat[z*stride : z*stride + N*K].view(N, K): ONE launch for the same Linear
    of every layer when its input is shared (r_net on the position table, model.py:278,313)."""
    M = A.shape[0]
    assert A.dtype == BF16 and Bflat.dtype == BF16 and A.shape[1] == K and Bflat.is_contiguous()
    assert Bflat.numel() >= (batch - 1) * stride + N * K
    if out is None:
        out = torch.empty(batch, M, N, device=A.device, dtype=BF16)
    call("commu_gemm_nt_bf16_batched", _p(A), _rowmajor2d(A, "A"), 0, _p(Bflat), K, stride, _p(out), N, M * N, M, N, K,
         None, 0, 0, 0, batch, 0, 0, _s())
    return out


def signbits_words(M, N, K, lda=None, ldb=None, ldc=None):
    """32-bit words of the sign-bit buffer of an M x N GEMM output (gemm_nt sign_bits_out / relu_bits); 0 when the shape
    does not take the one-bit ReLU mask (then the bf16 relu_mask is the path)."""
    return int(call("commu_gemm_nt_signbits_words", M, N, K, K if lda is None else lda, K if ldb is None else ldb,
                    N if ldc is None else ldc))


def gemm_nt(A, B, out=None, *, bias=None, resid=None, relu=False, relu_mask=None, out_f32=False,
            drop_p=0.0, drop_seed=0, mask_scale=1.0, sign_bits_out=None, relu_bits=None):
    """out[M,N] = A[M,K] @ B[N,K]^T with the fused epilogue bias -> relu -> dropout -> +resid ->
    relu-mask (kept values * mask_scale).  A, B bf16.
    sign_bits_out (int32 [signbits_words]): receives one bit per output, (out > 0); relu_bits: such a buffer from a GEMM
    with the same M x N, used instead of relu_mask (out = bit ? out * mask_scale : 0) -- 1/16 of the mask's bytes."""
    lda, ldb = _rowmajor2d(A, "A"), _rowmajor2d(B, "B")
    M, K = A.shape
    N = B.shape[0]
    assert B.shape[1] == K and A.dtype == BF16 and B.dtype == BF16
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=F32 if out_f32 else BF16)
    ldc = _rowmajor2d(out, "out")
    flags = 0
    if bias is not None:
        flags |= EPI_BIAS
        assert bias.dtype == F32 and bias.numel() >= N
    if resid is not None:
        flags |= EPI_RESID
        assert resid.dtype == BF16
    if relu:
        flags |= EPI_RELU
    if relu_mask is not None:
        flags |= EPI_RELUMASK
        assert relu_mask.dtype == BF16
    if out.dtype == F32:
        flags |= EPI_OUT_F32
    if drop_p > 0:
        flags |= EPI_DROPOUT
    if sign_bits_out is not None or relu_bits is not None:
        bits = sign_bits_out if sign_bits_out is not None else relu_bits
        assert relu_mask is None and not (sign_bits_out is not None and relu_bits is not None)
        assert bits.dtype == torch.int32 and bits.is_contiguous() and bits.numel() * 32 == M * N
        flags |= EPI_SIGNBITS_OUT if sign_bits_out is not None else EPI_RELUBITS
        relu_mask = bits          # (the C entry point takes the word buffer in the relu_mask argument)
        call("commu_gemm_nt_bf16", _p(A), lda, _p(B), ldb, _p(out), ldc, M, N, K, _p(bias), _p(resid),
             0 if resid is None else _rowmajor2d(resid, "resid"), _p(bits), 0,
             flags, int(drop_seed), float(drop_p), float(mask_scale), _s())
        return out
    call("commu_gemm_nt_bf16", _p(A), lda, _p(B), ldb, _p(out), ldc, M, N, K, _p(bias), _p(resid),
         0 if resid is None else _rowmajor2d(resid, "resid"), _p(relu_mask),
         0 if relu_mask is None else _rowmajor2d(relu_mask, "relu_mask"), flags, int(drop_seed), float(drop_p),
         float(mask_scale), _s())
    return out


def gemm_nt_ln(z, gamma, beta, B, out=None, *, a_out=None, bias=None, resid=None, relu=False, out_f32=False, eps=1e-5):
    """Decode-step Linear with the preceding LayerNorm fused in: out = LN(z) @ B^T (+ bias, relu, resid); `a_out`
    (bf16, same shape as z) receives LN(z).  z: [M <= 64, K] bf16 (K % 128 == 0); LN over gamma.numel() columns."""
    M, K = z.shape
    N = B.shape[0]
    assert z.dtype == BF16 and B.dtype == BF16 and B.shape[1] == K
    if out is None:
        out = torch.empty(M, N, device=z.device, dtype=F32 if out_f32 else BF16)
    if not FUSE_DECODE_LN or M > 64 or K % 128 or K > 1024 or N < 32:
        # shapes the fused kernel does not take (tiny test models, wide batches): the two kernels it replaces
        a, _, _ = layernorm_fwd(z, gamma, beta, y=a_out, eps=eps)
        return gemm_nt(a, B, out=out, bias=bias, resid=resid, relu=relu)
    flags = (EPI_BIAS if bias is not None else 0) | (EPI_RESID if resid is not None else 0) | (EPI_RELU if relu else 0)
    if out.dtype == F32:
        flags |= EPI_OUT_F32
    call("commu_gemm_nt_ln_bf16", _p(z), _rowmajor2d(z, "z"), _p(gamma), _p(beta), gamma.numel(), float(eps),
         _p(a_out), 0 if a_out is None else _rowmajor2d(a_out, "a_out"), _p(B), _rowmajor2d(B, "B"), _p(out),
         _rowmajor2d(out, "out"), M, N, K, _p(bias), _p(resid), 0 if resid is None else _rowmajor2d(resid, "resid"),
         flags, _s())
    return out


def tn_slices(M: int, N: int, K: int) -> int:
    """Number of m-slices the TN kernel wants at this shape (enough workgroups to fill the GPU)."""
    return _lib.load().commu_gemm_tn_slices(int(M), int(N), int(K))


def gemm_tn(A, B, out, *, accumulate=False, slabs=None, mode=None):
    """out[N,K] (+)= A[M,N]^T @ B[M,K]  (fp32 out; bf16 inputs).  Split over M into slabs that
    are summed by commu_reduce_slabs_f32.  `out` may be a strided 2-D view only if contiguous."""
    lda, ldb = _rowmajor2d(A, "A"), _rowmajor2d(B, "B")
    M, N = A.shape
    K = B.shape[1]
    assert B.shape[0] == M and out.shape == (N, K) and out.dtype == F32 and out.is_contiguous()
    ns = tn_slices(M, N, K)
    if slabs is None or slabs.numel() < ns * N * K:
        slabs = torch.empty(ns * N * K, device=A.device, dtype=F32)
    call("commu_gemm_tn_bf16", _p(A), lda, _p(B), ldb, _p(slabs), K, N * K, M, N, K, ns,
         TN_MODE if mode is None else mode, _s())
    call("commu_reduce_slabs_f32", _p(out), _p(slabs), N * K, ns, N * K, 1 if accumulate else 0, 1.0, _s())
    return out


def gemm_tn_raw(A, B, slabs, nslices, mode=None):
    """slabs[s][N,K] = A[m-slice s]^T @ B[m-slice s]; reduce with reduce_slabs."""
    lda, ldb = _rowmajor2d(A, "A"), _rowmajor2d(B, "B")
    M, N = A.shape
    K = B.shape[1]
    assert B.shape[0] == M and slabs.numel() >= nslices * N * K and slabs.dtype == F32
    call("commu_gemm_tn_bf16", _p(A), lda, _p(B), ldb, _p(slabs), K, N * K, M, N, K, nslices,
         TN_MODE if mode is None else mode, _s())
    return slabs


def tn_group(pairs, colsum=None):
    """ctypes problem array for a grouped weight-gradient launch: pairs = [(dY [M,N], X [M,K]), ...] (bf16, shared M).
    Returns (array, M, offsets, total) with offsets[i] = element offset of problem i's [N, K] block inside a slab.
    colsum: per problem, whether the launch also leaves the column sums of dY (the Linear's bias gradient) in the slabs;
    then a fifth value, the slab offset of each problem's [N] vector (None where not asked), is returned."""
    arr = (_lib.TnProblem * len(pairs))()
    M = pairs[0][0].shape[0]
    offs, total = [], 0
    for i, (A, B) in enumerate(pairs):
        assert A.shape[0] == M and B.shape[0] == M and A.dtype == BF16 and B.dtype == BF16
        arr[i].A, arr[i].B = A.data_ptr(), B.data_ptr()
        arr[i].lda, arr[i].ldb = _rowmajor2d(A, "A"), _rowmajor2d(B, "B")
        arr[i].N, arr[i].K = A.shape[1], B.shape[1]
        arr[i].out_off = total
        arr[i].colsum_off = -1
        offs.append(total)
        total += A.shape[1] * B.shape[1]
    if colsum is None:
        return arr, M, offs, total
    cs = []
    for i, want in enumerate(colsum):
        if want:
            arr[i].colsum_off = total
            cs.append(total)
            total += round_up(pairs[i][0].shape[1], 4)
        else:
            cs.append(None)
    return arr, M, offs, total, cs


def tn_group_slices(arr, M, budget=None):
    """Token slices the grouped kernel wants for this group (0: not eligible -> per-problem commu_gemm_tn_bf16).
    budget: workgroups the launch may take (default: the library's 128 = half of every XCD, for a launch beside the backward pass)."""
    if budget is not None:
        return _lib.load().commu_gemm_tn_grouped_slices_budget(arr, len(arr), int(M), int(budget))
    return _lib.load().commu_gemm_tn_grouped_slices(arr, len(arr), int(M))


def gemm_tn_grouped(arr, M, slabs, slab_stride, nslices):
    """slabs[s * slab_stride + off_p + n*K_p + k] = sum over token slice s of A_p[m,n] B_p[m,k], one launch."""
    if not slabs.is_cuda:
        raise CommuHipError("commu_amd kernels need GPU tensors (no CPU fallback)")
    assert slabs.dtype == F32 and slabs.numel() >= nslices * slab_stride
    call("commu_gemm_tn_bf16_grouped", arr, len(arr), int(M), _p(slabs), int(slab_stride), int(nslices), _s())
    return slabs


def reduce_slabs(dst, slabs, n, nslabs, stride, accumulate, alpha=1.0):
    """dst.view(-1)[:n] = (accumulate ? dst : 0) + alpha * sum_s slabs[s*stride : s*stride + n]"""
    assert dst.is_contiguous() and dst.dtype == F32 and dst.numel() >= n
    call("commu_reduce_slabs_f32", _p(dst), _p(slabs), n, nslabs, stride, 1 if accumulate else 0, float(alpha), _s())
    return dst


def quant_mxfp8(X):
    """bf16 [rows, K] -> (e4m3 bytes uint8 [rows, K], E8M0 scale bytes uint8 [rows, K/32])  (OCP MX, block 32)."""
    rows, K = X.shape
    assert X.dtype == BF16 and X.stride(1) == 1 and K % 32 == 0
    Q = torch.empty(rows, K, device=X.device, dtype=torch.uint8)
    S = torch.empty(rows, K // 32, device=X.device, dtype=torch.uint8)
    call("commu_quant_mxfp8", _p(X), X.stride(0), _p(Q), K, _p(S), K // 32, rows, K, _s())
    return Q, S


def gemm_nt_mxfp8(A, SA, B, SB, out=None, bias=None, relu=False, resid=None, drop_p=0.0, drop_seed=0):
    """out bf16 [M, N] = A . B^T with MX-fp8 operands from quant_mxfp8 (A [M, K], B [N, K]); fp32 accumulation; the
    epilogue (bias -> ReLU -> dropout -> residual) is commu_gemm_nt_bf16's."""
    M, K = A.shape
    N = B.shape[0]
    assert A.dtype == torch.uint8 and B.dtype == torch.uint8 and B.shape[1] == K
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=BF16)
    flags = (EPI_BIAS if bias is not None else 0) | (EPI_RELU if relu else 0) | (EPI_RESID if resid is not None else 0) | \
            (EPI_DROPOUT if drop_p > 0 else 0)
    call("commu_gemm_nt_mxfp8", _p(A), A.stride(0), _p(SA), SA.stride(0), _p(B), B.stride(0), _p(SB), SB.stride(0), _p(out),
         out.stride(0), M, N, K, _p(bias), _p(resid), 0 if resid is None else resid.stride(0), flags, int(drop_seed),
         float(drop_p), _s())
    return out


def linear_mxfp8(x, wq, out=None, **epi):
    """x bf16 [M, K] quantised on the fly, wq = (bytes, scales) of the weight: one Linear in MX-fp8."""
    xq, xs = quant_mxfp8(x)
    return gemm_nt_mxfp8(xq, xs, wq[0], wq[1], out=out, **epi)


def reduce_slabs_group(items, slabs, nslabs, stride, accumulate, alpha=1.0):
    """items = [(dst fp32 contiguous, src_off, crop (rg, rt, rp, cg, ct, cp))]: every reduction of a grouped
    weight-gradient launch in ONE launch (<= 8 items)."""
    arr = (_lib.ReduceItem * len(items))()
    for i, (dst, off, crop) in enumerate(items):
        rg, rt, rp, cg, ct, cp = crop
        assert dst.is_contiguous() and dst.dtype == F32 and dst.numel() >= rg * rt * cg * ct
        arr[i].dst, arr[i].src_off = dst.data_ptr(), off
        arr[i].rg, arr[i].rt, arr[i].rp, arr[i].cg, arr[i].ct, arr[i].cp = rg, rt, rp, cg, ct, cp
    call("commu_reduce_slabs_group_f32", arr, len(items), _p(slabs), nslabs, stride, 1 if accumulate else 0, float(alpha), _s())


def reduce_slabs_crop(dst, slabs, crop, nslabs, stride, accumulate, alpha=1.0):
    """crop = (rg, rt, rp, cg, ct, cp): the [rt, ct] blocks of the padded [rg*rp, cg*cp] product (summed over the
    slabs) are added to / stored in dst [rg*rt, cg*ct]."""
    rg, rt, rp, cg, ct, cp = crop
    assert dst.is_contiguous() and dst.dtype == F32 and dst.numel() == rg * rt * cg * ct
    call("commu_reduce_slabs_crop_f32", _p(dst), _p(slabs), rg, rt, rp, cg, ct, cp, nslabs, stride,
         1 if accumulate else 0, float(alpha), _s())
    return dst


def embed_fwd(tok, E, out=None, drop_p=0.0, drop_seed=0, ld=None):
    """ld > D: rows are padded to ld columns (zero-padding contract of commu_hip.h)."""
    ntok = tok.numel()
    D = E.shape[1]
    assert tok.dtype == torch.int64 and E.dtype == F32 and E.is_contiguous() and tok.is_contiguous()
    if out is None:
        out = torch.empty(ntok, D if ld is None else ld, device=E.device, dtype=BF16)
    call("commu_embed_fwd", _p(tok), _p(E), _p(out), out.stride(0), ntok, D, E.shape[0], math.sqrt(D), int(drop_seed),
         float(drop_p), _s())
    return out


def token_order(tok, V):
    """(perm, offs) for embed_bwd: stable argsort of the ids and the first sorted position of every id (int64 [V+1]).
    V <= 1024: the library's counting sort (three small launches, no framework kernels); ids outside [0, V) are left out
    of the order (they contribute nothing either way).  Larger vocabularies: torch.sort + searchsorted."""
    flat = tok.reshape(-1)
    if V <= 1024 and flat.is_cuda and flat.dtype == torch.int64 and flat.is_contiguous():
        n = flat.numel()
        perm = torch.empty(n, device=flat.device, dtype=torch.int64)
        offs = torch.empty(V + 1, device=flat.device, dtype=torch.int64)
        ws = torch.empty(64 * V, device=flat.device, dtype=torch.int32)
        call("commu_token_order", _p(flat), n, V, _p(perm), _p(offs), _p(ws), _s())
        return perm, offs
    vals, perm = torch.sort(flat, stable=True)
    offs = torch.searchsorted(vals, torch.arange(V + 1, device=tok.device, dtype=vals.dtype))
    return perm, offs


def embed_bwd(tok, dX, dE, accumulate=True, drop_p=0.0, drop_seed=0, order=None):
    """dE[v] (+)= sqrt(D) * sum over the tokens with id v of the (dropout-masked) dX rows.  order = token_order(tok, V):
    the sorted two-pass kernel (work per wave independent of how often an id occurs, fixed summation order); None: one
    workgroup per row scans the token list."""
    V, D = dE.shape
    if order is not None and dE.is_contiguous():
        perm, offs = order
        ntok = perm.numel()
        ws = torch.empty(call("commu_embed_bwd_ws_rows", ntok, V), D, device=dE.device, dtype=F32)
        call("commu_embed_bwd_sorted", _p(perm), _p(offs), _p(dX), dX.stride(0), _p(ws), ntok, D, V, _p(dE), math.sqrt(D),
             1 if accumulate else 0, int(drop_seed), float(drop_p), _s())
        return dE
    call("commu_embed_bwd", _p(tok), _p(dX), dX.stride(0), _p(dE), tok.numel(), D, V, math.sqrt(D),
         1 if accumulate else 0, int(drop_seed), float(drop_p), _s())
    return dE


def posemb(inv_freq, K, D, out=None, drop_p=0.0, drop_seed=0, ld=None, clamp_len=-1):
    """Sinusoid table by distance; clamp_len > 0: distances above it share its row (cfg.MODEL.clamp_len, model.py:581-582)."""
    if out is None:
        out = torch.empty(K, D if ld is None else ld, device=inv_freq.device, dtype=BF16)
    call("commu_posemb_fwd", _p(inv_freq), _p(out), out.stride(0), K, D, int(clamp_len), int(drop_seed), float(drop_p), _s())
    return out


def layernorm_fwd(z, gamma, beta, y=None, mean=None, rstd=None, eps=1e-5, y_drop=None, drop_p=0.0, drop_seed=0):
    """y = LN(z) over the first D = gamma.numel() columns (wider rows: zero-padding contract); optionally also
    y_drop = dropout(y) (second output)."""
    rows, D = z.shape[0], gamma.numel()
    y = torch.empty_like(z) if y is None else y
    mean = torch.empty(rows, device=z.device, dtype=F32) if mean is None else mean
    rstd = torch.empty(rows, device=z.device, dtype=F32) if rstd is None else rstd
    call("commu_layernorm_fwd", _p(z), z.stride(0), _p(gamma), _p(beta), _p(y), y.stride(0), _p(mean), _p(rstd),
         rows, D, eps, _p(y_drop), 0 if y_drop is None else y_drop.stride(0), int(drop_seed), float(drop_p), _s())
    return y, mean, rstd


def layernorm_bwd(dy, z, mean, rstd, gamma, dz=None, part=None, dz_masked=None, drop_p=0.0, drop_seed=0, add=None):
    """Returns (dz, part) with part [nblk, 3, D]: partial column sums of dy*xhat, dy, dz (or of
    dz_masked = dz * keep/(1-p) when that second output is requested).  D = gamma.numel().
    add (optional, bf16 like dy): the incoming gradient is dy + add (a residual branch's gradient)."""
    rows, D = z.shape[0], gamma.numel()
    nblk = call("commu_layernorm_bwd_nblocks", rows)
    dz = torch.empty_like(z) if dz is None else dz
    if part is None:
        part = torch.empty(nblk, 3, D, device=z.device, dtype=F32)
    if add is not None:
        call("commu_layernorm_bwd_add", _p(dy), dy.stride(0), _p(add), add.stride(0), _p(z), z.stride(0), _p(mean), _p(rstd),
             _p(gamma), _p(dz), dz.stride(0), _p(part), rows, D, _p(dz_masked), 0 if dz_masked is None else dz_masked.stride(0),
             int(drop_seed), float(drop_p), _s())
        return dz, part[:nblk]
    call("commu_layernorm_bwd", _p(dy), dy.stride(0), _p(z), z.stride(0), _p(mean), _p(rstd), _p(gamma), _p(dz),
         dz.stride(0), _p(part), rows, D, _p(dz_masked), 0 if dz_masked is None else dz_masked.stride(0),
         int(drop_seed), float(drop_p), _s())
    return dz, part[:nblk]


def gelu_fwd(z, drop_p=0.0, drop_seed=0, out=None):
    """out = dropout(gelu(z)) (exact erf form); z bf16 2-D.  Non-default FFN activation (commu_hip.h)."""
    rows, cols = z.shape
    if out is None:
        out = torch.empty(rows, z.stride(0), device=z.device, dtype=BF16)[:, :cols]
    call("commu_gelu_fwd", _p(z), z.stride(0), _p(out), out.stride(0), rows, cols, int(drop_seed), float(drop_p), _s())
    return out


def gelu_bwd(dy, z, drop_p=0.0, drop_seed=0, out=None):
    """dz = dy * keep/(1-p) * gelu'(z)."""
    rows, cols = z.shape
    if out is None:
        out = torch.empty(rows, z.stride(0), device=z.device, dtype=BF16)[:, :cols]
    call("commu_gelu_bwd", _p(dy), dy.stride(0), _p(z), z.stride(0), _p(out), out.stride(0), rows, cols, int(drop_seed),
         float(drop_p), _s())
    return out


class ColsumGroup:
    """Collects the final column-sum passes of a backward step (bias, LayerNorm-parameter and shared attention-bias
    gradients) and runs them as ONE launch (commu_colsum_group_f32): sources that add into the same output are walked by
    the same workgroups, in the order they were added.  colsum(..., group=g) / layernorm_bwd_reduce(..., group=g) only run
    their slab pass (if any) right away; flush() launches the rest on the current stream."""

    def __init__(self):
        self.by_out = {}          # out.data_ptr() -> [out, cols, [(X tensor, ptr, ldx, rows, alpha), ...]]
        self.keep = []

    def add(self, out, cols, X, ptr, ldx, rows, alpha):
        ent = self.by_out.setdefault(out.data_ptr(), [out, cols, []])
        assert ent[1] == cols
        ent[2].append((ptr, ldx, rows, float(alpha)))
        self.keep.append(X)

    def flush(self):
        # (no record_stream on the sources: the caller keeps this object -- and with it every source -- alive until the
        #  streams are joined; recording ~60 cross-stream uses per step made every later allocation poll their events:
        #  the host-bound step at 8 sequences per GPU went from 5.4 to 10-13 ms)
        tasks, srcs = [], []

        def launch():
            if not tasks:
                return
            ta = (_lib.ColsumTask * len(tasks))()
            sa = (_lib.ColsumSource * len(srcs))()
            for i, (o, cols, b, e) in enumerate(tasks):
                ta[i].out, ta[i].cols, ta[i].src_begin, ta[i].src_end = o, cols, b, e
            for i, (ptr, ldx, rows, alpha) in enumerate(srcs):
                sa[i].X, sa[i].ldx, sa[i].rows, sa[i].alpha = ptr, ldx, rows, alpha
            call("commu_colsum_group_f32", ta, len(tasks), sa, len(srcs), _s())
            tasks.clear()
            srcs.clear()
        for out, cols, lst in self.by_out.values():
            if len(tasks) + 1 > 48 or len(srcs) + len(lst) > 64:
                launch()
            b = len(srcs)
            srcs.extend(lst)
            tasks.append((out.data_ptr(), cols, b, len(srcs)))
        launch()
        self.by_out = {}          # (self.keep stays: see above)


def layernorm_bwd_reduce(part, dgamma, dbeta, dbias=None, group=None):
    """One launch for the three partial-sum planes of layernorm_bwd: accumulates into the gradients (group: deferred
    into the step's grouped final pass)."""
    nblk, _, D = part.shape
    assert part.is_contiguous() and part.dtype == F32
    if group is not None:
        for z, dst in enumerate((dgamma, dbeta, dbias)):
            if dst is not None:
                group.add(dst, D, part, part.data_ptr() + 4 * z * D, 3 * D, nblk, 1.0)
        return
    call("commu_layernorm_bwd_reduce", _p(part), nblk, D, _p(dgamma), _p(dbeta), _p(dbias), _s())


def colsum(X, out, alpha=1.0, group=None):
    """out[c] += alpha * sum_r X[r, c]  (X bf16 or fp32; deterministic two-pass sum, no atomics).  group: only the slab
    pass runs now, the final pass joins the group's single launch."""
    rows, cols = X.shape
    bf = X.dtype == BF16
    ny = call("commu_colsum_slabs", rows, cols, 2 if bf else 4)
    ws = torch.empty(ny, round_up(cols, 8 if bf else 4), device=X.device, dtype=F32) if ny > 0 else None
    if group is not None:
        if ny > 0:
            got = call("commu_colsum_slab_pass", _p(X), 2 if bf else 4, X.stride(0), rows, cols, _p(ws), ny, _s())
            if got != ny:
                raise CommuHipError(f"commu_colsum_slab_pass returned {got} (expected {ny})")
            group.add(out, cols, ws, ws.data_ptr(), ws.stride(0), ny, alpha)
        else:
            assert X.dtype == F32
            group.add(out, cols, X, X.data_ptr(), X.stride(0), rows, alpha)
        return out
    call("commu_colsum_bf16" if bf else "commu_colsum_f32", _p(X), X.stride(0), rows, cols, _p(out), _p(ws), ny,
         float(alpha), _s())
    return out


def ce_fwd(logits, target, V):
    rows = logits.shape[0]
    nll = torch.empty(rows, device=logits.device, dtype=F32)
    lse = torch.empty(rows, device=logits.device, dtype=F32)
    call("commu_ce_fwd", _p(logits), logits.stride(0), _p(target), _p(nll), _p(lse), rows, V, _s())
    return nll, lse


def ce_bwd(logits, target, lse, g, V, dlogits=None):
    rows = logits.shape[0]
    if dlogits is None:
        dlogits = torch.empty(rows, logits.stride(0), device=logits.device, dtype=BF16)
    call("commu_ce_bwd", _p(logits), logits.stride(0), _p(target), _p(lse), _p(g), _p(dlogits), dlogits.stride(0),
         rows, V, _s())
    return dlogits


def mems_update(hids, mems, out, beg):
    """K9: out[l] = cat(mems[l], hids[l])[beg:]  with hids [L+1, T*B, Dp], mems [L+1, M, B, Dp] (layer stride free),
    out [L+1, n, B, Dp] contiguous, 0 <= beg < M."""
    Lp, _, B, Dp = out.shape
    M = mems.shape[1]
    step = B * Dp
    assert 0 <= beg < M and hids.shape[0] == Lp == mems.shape[0] and out.is_contiguous()
    assert out.shape[1] * step == (M - beg) * step + hids.shape[1] * Dp
    call("commu_mems_update", _p(hids), hids.stride(0), 0, hids.shape[1] * Dp, _p(mems), mems.stride(0), beg * step,
         (M - beg) * step, _p(out), out.stride(0), Lp, _s())
    return out


def masked_mean(nll, target, pad, scale, ws_sum, ws_cnt, out):
    call("commu_masked_mean", _p(nll), _p(target), nll.numel(), pad, scale, _p(ws_sum), _p(ws_cnt), _p(out), _s())
    return out


def loss_grad(target, pad, ws_cnt, scale, g):
    call("commu_loss_grad", _p(target), target.numel(), pad, _p(ws_cnt), scale, _p(g), _s())
    return g


def masked_mean_groups(nll, target, pad, scale, B, Bc, ws_sum, ws_cnt, out, sum_all=None):
    """[T, B] loss whose columns are B / Bc micro-batches: out = scale * sum over the groups of their masked means."""
    call("commu_masked_mean_groups", _p(nll), _p(target), nll.numel(), pad, scale, B, Bc, _p(ws_sum), _p(ws_cnt), _p(out),
         _p(sum_all), _s())
    return out


def loss_grad_groups(target, pad, ws_cnt, scale, B, Bc, g):
    call("commu_loss_grad_groups", _p(target), target.numel(), pad, _p(ws_cnt), scale, B, Bc, _p(g), _s())
    return g


def grad_norm(g, part, out):
    call("commu_grad_norm", _p(g), g.numel(), _p(part), part.numel(), _p(out), _s())
    return out


def adam_step(p, g, m, v, p_bf16, lr, step, gnorm=None, clip=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
    call("commu_adam_step", _p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), lr, beta1, beta2, eps, step,
         _p(gnorm), clip, _s())


def adam_step_dev(p, g, m, v, p_bf16, scal, gnorm=None, clip=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
    """adam_step with {lr, 1 - beta1^t, 1 - beta2^t} read from the device tensor `scal` (fp32 [>= 3]): graph-replayable."""
    assert scal.dtype == F32 and scal.numel() >= 3
    call("commu_adam_step_dev", _p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), _p(scal), beta1, beta2, eps,
         _p(gnorm), clip, _s())


def adam_bias_corrections(beta1, beta2, step):
    """(1 - beta1^step, 1 - beta2^step) in the library's own float arithmetic (bit-identical to commu_adam_step's)."""
    out = (C.c_float * 2)()
    call("commu_adam_bias_corrections", float(beta1), float(beta2), int(step), out)
    return float(out[0]), float(out[1])


def set_seed_salt(src):
    """Every dropout site adds the uint32 at device tensor `src` (int32 [1]) to its seed from now on (None: back to 0)."""
    call("commu_set_seed_salt", _p(src), _s())


def scale_clip(g, gnorm, clip):
    """g *= min(1, clip / (gnorm + 1e-6))   (torch.nn.utils.clip_grad_norm_, train.py:159-161)"""
    call("commu_scale_clip_f32", _p(g), g.numel(), _p(gnorm), clip, _s())
    return g


def cast_bf16(x, out=None):
    out = torch.empty(x.shape, device=x.device, dtype=BF16) if out is None else out
    call("commu_cast_f32_bf16", _p(x), _p(out), x.numel(), _s())
    return out


def cast_f32(x, out=None):
    out = torch.empty(x.shape, device=x.device, dtype=F32) if out is None else out
    call("commu_cast_bf16_f32", _p(x), _p(out), x.numel(), _s())
    return out


def transpose_to_bf16(x, out=None):
    """out[c, r] = bf16(x[r, c]) for a 2-D fp32 or bf16 x."""
    rows, cols = x.shape
    out = torch.empty(cols, rows, device=x.device, dtype=BF16) if out is None else out
    name = "commu_transpose_f32_bf16" if x.dtype == F32 else "commu_transpose_bf16"
    call(name, _p(x), x.stride(0), _p(out), out.stride(0), rows, cols, _s())
    return out


def transpose_group_bf16(pairs):
    """out[c, r] = x[r, c] for every (x, out) pair of bf16 2-D tensors, 32 pairs per launch."""
    for i in range(0, len(pairs), 32):
        part = pairs[i:i + 32]
        arr = (_lib.TransposeItem * len(part))()
        for k, (x, out) in enumerate(part):
            if not x.is_cuda:
                raise CommuHipError("commu_amd kernels need GPU tensors (no CPU fallback)")
            assert x.dtype == BF16 and out.dtype == BF16 and x.stride(1) == 1 and out.stride(1) == 1
            rows, cols = x.shape
            assert out.shape[0] == cols and out.shape[1] >= rows
            arr[k].src, arr[k].dst = x.data_ptr(), out.data_ptr()
            arr[k].ldi, arr[k].ldo, arr[k].rows, arr[k].cols = x.stride(0), out.stride(0), rows, cols
        call("commu_transpose_group_bf16", arr, len(part), _s())


def transpose_heads(src, J, B, H, DH, W, off=0, bias=None, out=None):
    """out[b,h,f, off+j] = src[(j*B+b), h*DH+f] (+bias[h*DH+f]); other columns zero.  src is a 2-D view."""
    if out is None:
        out = torch.empty(B, H, DH, W, device=src.device, dtype=BF16)
    call("commu_transpose_heads", _p(src), src.stride(0), _p(bias), _p(out), J, B, H, DH, W, off, _s())
    return out


def round_up(x, m):
    return (x + m - 1) // m * m


def _attn_desc(q, k, v, rd, u, vb, reset, T, M, B, H, DH, ld_o, same_length, mem_len, drop_p=0.0, drop_seed=0,
               scale=None):
    d = AttnDesc()
    d.drop_p, d.drop_seed = float(drop_p), int(drop_seed)
    d.q, d.k, d.v, d.rd = q.data_ptr(), k.data_ptr(), v.data_ptr(), rd.data_ptr()
    d.r_w_bias, d.r_r_bias = u.data_ptr(), vb.data_ptr()
    d.reset = reset.data_ptr() if reset is not None else None
    d.ld_qkv, d.ld_rd, d.ld_o = q.stride(0), rd.stride(0), ld_o
    d.T, d.M, d.B, d.H, d.DH = T, M, B, H, DH
    K = T + M
    # model.py:549-568: mask_len = klen - mem_len; shift = qlen - mask_len if mask_len > 0 else qlen
    d.same_length = 1 if same_length else 0
    mask_len = K - mem_len
    d.sshift = (T - mask_len) if mask_len > 0 else T
    # (scale: 1/sqrt(true d_head) when the head dimension is zero-padded, e.g. 50 -> 64)
    d.scale = 1.0 / math.sqrt(DH) if scale is None else float(scale)
    return d


# Training forward (d_head 64) that SAVES its probabilities for the backward pass (commu_relattn_fwd_save): the
# query-stationary backward kernel then recomputes neither scores, rel-shift, masks, exponentials nor the dropout hash (2 of
# its 4 products remain).  Built, tested (tests/test_kernels_gpu.py "forward_p") and OFF by default: on MI355X the step is
# bound by memory traffic, and the trade buys compute with bytes -- relattn_bwd_q 690 -> 645 us, relattn_fwd3 297 -> 358 us
# (+0.59 GB of stores per launch through a store path that is already the forward's second-largest cost), the step within
# +-1 % (16.48 -> 16.36 ms on one box), +6.8 GB of HBM traffic and +6.4 GB of live memory per step.  Feeding the
# key-stationary kernel from the same tiles (so that no kernel stores P) was built too: that kernel 292 -> 387 us, the step
# 15.80 -> 16.10 ms.  DESIGN.md section 8e.
import os as _os
FWD_SAVES_P = _os.environ.get("COMMU_FWD_SAVES_P", "0") == "1"


def relattn_fwd(q, k, v, rd, u, vb, reset, T, M, B, H, DH, same_length, mem_len, out=None, lse=None,
                save_q=False, drop_p=0.0, drop_seed=0, scale=None, save_p=False):
    """q: 2-D view [T*B, H*DH] (row stride ld_qkv), k, v: [(T+M)*B, H*DH]; rd: [K, H*DH] by distance.
    Returns (out bf16 [T*B, H*DH], lse fp32 [B,H,T], (qu2, qv2) or None); with save_p (d_head 64, generation-3 forward)
    the third value is (qu2, qv2, pf): pf = the forward pass's probabilities for relattn_bwd (None when not available)."""
    if out is None:
        out = torch.empty(T * B, H * DH, device=q.device, dtype=BF16)
    if lse is None:
        lse = torch.empty(B, H, T, device=q.device, dtype=F32)
    qs = None
    if save_q:
        qs = (torch.empty(T * B, H * DH, device=q.device, dtype=BF16), torch.empty(T * B, H * DH, device=q.device, dtype=BF16))
    d = _attn_desc(q, k, v, rd, u, vb, reset, T, M, B, H, DH, out.stride(0), same_length, mem_len, drop_p, drop_seed,
                   scale)
    if save_p and qs is not None:
        pf = None
        if DH == 64 and FWD_SAVES_P and call("commu_attn_fwd_generation", -1) in (0, 3):
            pf = torch.empty(call("commu_attn_pf_bytes", T, M, B, H), device=q.device, dtype=torch.uint8)
            if POISON_SCRATCH:
                pf.fill_(0xFF)          # (bf16 / fp32 NaN patterns: tiles the forward never writes must never be used)
            call("commu_relattn_fwd_save", C.byref(d), _p(out), _p(lse), _p(qs[0]), _p(qs[1]), _p(pf), _s())
            return out, lse, (qs[0], qs[1], pf)
        call("commu_relattn_fwd", C.byref(d), _p(out), _p(lse), _p(qs[0]), _p(qs[1]), _s())
        return out, lse, (qs[0], qs[1], None)
    call("commu_relattn_fwd", C.byref(d), _p(out), _p(lse), _p(qs[0]) if qs else None, _p(qs[1]) if qs else None, _s())
    return out, lse, qs


POISON_SCRATCH = False      # tests: fill uninitialised scratch with NaN to prove nothing reads it
# decode step: LayerNorm inside the consuming Linear (commu_gemm_nt_ln_bf16).  Measured on MI355X (64 sequences, L6
# D512): 0.324 ms / step fused vs 0.293 ms with the separate LayerNorm kernel -- the two LDS reductions and the
# normalisation arithmetic cost a latency-bound skinny GEMM more than the launch they save -- so it is OFF by default.
FUSE_DECODE_LN = False
STORE_ATTN_P = True         # backward: bwd_q stores P for bwd_kv (d_head 64); False: both kernels recompute it
# delta = rowsum(o . dO): the separate commu_attn_delta launch (True, the default) or inside the query-stationary kernel
# (commu_attn_bwd_desc.o).  In the step the two take the same time (round 4: 16.35 / 16.44 vs 16.36 / 16.44 ms; round 5, three
# interleaved pairs: 15.92-15.95 ms both): in-kernel, the 28 us launch and 0.35 GB of HBM traffic per step go (o is read once
# where the launch reads o and dO), but every one of bwd_q's 8192 short-lived workgroups pays a longer prologue (0.588 against
# 0.558 ms per launch in the step).  Equal: the kernel stays lean and the launch stays (COMMU_DELTA_KERNEL=0 selects the other).
DELTA_KERNEL = _os.environ.get("COMMU_DELTA_KERNEL", "1") != "0"
NO_FUSED_BAND = False       # tests / A-B runs: keep the two band GEMMs instead of commu_relattn_bwd_band


def relattn_bwd(q, k, v, rd, u, vb, reset, T, M, B, H, DH, same_length, mem_len, o, dout, lse, qs, dq, dk, dv,
                drd, du, dvb, drop_p=0.0, drop_seed=0, scale=None, scratch=None, defer=None, colsum_group=None,
                dq_colsum=True):
    """Backward of relattn_fwd.  dq/dk/dv: bf16 2-D views (row stride ld_dqkv) written in place;
    drd: fp32 [K, H*DH] (overwritten); du, dvb: fp32 [H*DH] accumulated into.  dq_colsum=False: the caller adds the
    colsum(dq) term of dvb itself (the model takes it from the qkv weight-gradient launch)."""
    dev = q.device
    K = T + M
    HD = H * DH
    qrows = call("commu_attn_bwd_qrows", T)
    QT = (T + qrows - 1) // qrows
    qu2, qv2 = qs[0], qs[1]
    pf = qs[2] if len(qs) > 2 else None          # probabilities saved by the forward pass (relattn_fwd(save_p=True))
    # delta[b,h,i] = sum_d o . dout: computed (and written, for the key-stationary kernel) by the query-stationary kernel
    # from the dout fragments it holds anyway (commu_attn_bwd_desc.o); DELTA_KERNEL: the separate launch instead
    delta = torch.empty(B, H, T, device=dev, dtype=F32)
    if DELTA_KERNEL or o.stride(0) != dout.stride(0):
        call("commu_attn_delta", _p(o), _p(dout), o.stride(0), _p(delta), T, B, H, DH, _s())
    # fused band pass (commu_relattn_bwd_band: dq_BD and dRd in ONE sweep over dS-by-distance) when the shape allows
    band_slabs = call("commu_attn_band_pairs", T, B, H) if (DH == 64 and K <= 4096 and not NO_FUSED_BAND) else 0
    ld_dsk = round_up(K, 128) if band_slabs else round_up(K, 32)
    # dS by distance is lower-triangular (d <= i + M).  Without same_length / reset masks the two GEMMs below only
    # visit the band (half the work) and the kernel only writes the triangle.  What lies right of the causal edge
    # must read as zero: either the kernel also writes a wedge of zeros as wide as a GEMM tile can overhang
    # (fresh, uninitialised scratch) or the caller keeps ONE zero-initialised scratch per shape alive
    # (`scratch` dict, used by the model): nothing ever writes beyond the triangle, so it stays zero -- no 1 GB
    # fill per layer and no wedge.
    # (reset_mems: the kernel writes zeros over the distances of the memory tiles a fresh sequence skips, so the band
    #  structure -- and the persistent scratch -- also hold for the training configurations with XL memory)
    band = not same_length
    key = (T, M, B, H, ld_dsk, str(dev))
    if band and scratch is not None and not POISON_SCRATCH:
        if scratch.get("key") != key:
            scratch["key"], scratch["dsk"] = key, None          # drop the old buffer before allocating the new one
            scratch["dsk"] = torch.zeros(H, T * B, ld_dsk, device=dev, dtype=BF16)
        dsk, wedge = scratch["dsk"], 0
    elif band:
        # (zeros right of the causal edge, as wide as a GEMM tile / a 256-distance chunk of the fused pass overhangs)
        wedge = (136 + (64 + B - 1) // B) if band_slabs else (136 + (128 + B - 1) // B)
        dsk = torch.empty(H, T * B, ld_dsk, device=dev, dtype=BF16)
        if POISON_SCRATCH:
            dsk.fill_(float("nan"))
    else:
        wedge = 0
        dsk = torch.zeros(H, T * B, ld_dsk, device=dev, dtype=BF16)
    tri_B, tri_M = (B, M) if band else (0, 0)
    dq_ac = torch.empty(T * B, HD, device=dev, dtype=BF16)
    du_part = torch.empty(B * QT, HD, device=dev, dtype=F32)
    d = _attn_desc(q, k, v, rd, u, vb, reset, T, M, B, H, DH, o.stride(0), same_length, mem_len, drop_p, drop_seed,
                   scale)
    e = AttnBwdDesc()
    e.dout, e.lse, e.delta = dout.data_ptr(), lse.data_ptr(), delta.data_ptr()
    e.qu2, e.qv2 = qu2.data_ptr(), qv2.data_ptr()
    e.dq_ac, e.dk, e.dv = dq_ac.data_ptr(), dk.data_ptr(), dv.data_ptr()
    e.dsk, e.du_part = dsk.data_ptr(), du_part.data_ptr()
    e.ld_dqkv, e.ld_dsk, e.du_rows, e.dsk_wedge = dk.stride(0), ld_dsk, QT, wedge
    e.dsk_tiled = 1 if band_slabs else 0
    pscr = None
    if DH == 64 and STORE_ATTN_P:
        # the query-stationary kernel stores the probabilities it recomputes, the key-stationary one reads them back
        n = call("commu_attn_p_scratch_elems", T, M, B, H)
        if scratch is not None and not POISON_SCRATCH:
            if scratch.get("p") is None or scratch["p"].numel() < n:
                scratch["p"] = None
                scratch["p"] = torch.empty(n, device=dev, dtype=BF16)
            pscr = scratch["p"]
        else:
            pscr = torch.empty(n, device=dev, dtype=BF16)
            if POISON_SCRATCH:
                pscr.fill_(float("nan"))
        e.p_scratch = pscr.data_ptr()
        if pf is not None:
            e.pf = pf.data_ptr()
    assert dv.stride(0) == dk.stride(0)
    if not (DELTA_KERNEL or o.stride(0) != dout.stride(0)):
        e.o = o.data_ptr()
    call("commu_relattn_bwd_q", C.byref(d), C.byref(e), _s())
    call("commu_relattn_bwd_kv", C.byref(d), C.byref(e), _s())
    c2 = d.scale * 1.4426950408889634
    TB = T * B
    if band_slabs:
        # one pass: dq = dq_ac + dsk . Rd on this stream; the dRd partial slabs are reduced off the critical path
        kpad = round_up(K, 512)          # slab rows: the kernel works in passes of 512 distances
        slabs = torch.empty(H * band_slabs * kpad * DH, device=dev, dtype=F32)
        call("commu_relattn_bwd_band", _p(dsk), ld_dsk, _p(rd), rd.stride(0), _p(qv2), qv2.stride(0), _p(dq_ac), HD,
             _p(dq), dq.stride(0), _p(slabs), T, M, B, H, DH, 1 if band else 0, _s())

        def drd_reduce():
            if defer is not None:
                slabs.record_stream(torch.cuda.current_stream())
            call("commu_reduce_slabs2d_f32", _p(drd), drd.stride(0), DH, _p(slabs), K, DH, band_slabs, kpad * DH, H, 0,
                 1.0 / c2, _s())
        if defer is None:
            drd_reduce()
        else:
            defer(drd_reduce)
        _bias_grads(dq if dq_colsum else None, du_part, du, dvb, HD, dev, defer, colsum_group)
        return delta
    # BD part of dq and dRd: two GEMMs per head over dS-by-distance, batched over the heads
    rdt = transpose_heads(rd, K, 1, H, DH, ld_dsk)                   # [1, H, DH, ld_dsk]
    call("commu_gemm_nt_bf16_batched", _p(dsk), ld_dsk, TB * ld_dsk, _p(rdt), ld_dsk, DH * ld_dsk, _p(dq), dq.stride(0),
         DH, TB, DH, ld_dsk, _p(dq_ac), HD, DH, EPI_RESID, H, tri_B, tri_M, _s())
    def drd_part():
        # dRd only feeds r_net's weight gradient: `defer` (optional) runs it off the critical path (side stream)
        if defer is not None:
            # a call-local dS-by-distance buffer (reset / same_length masks, or no persistent scratch) must outlive
            # this function on the deferring stream: without this the main-stream allocator may hand the block to
            # the next layer while the GEMM below is still reading it
            dsk.record_stream(torch.cuda.current_stream())
            qv2.record_stream(torch.cuda.current_stream())
        ns = tn_slices(TB, ld_dsk, DH * H)
        slabs = torch.empty(H * ns * ld_dsk * DH, device=dev, dtype=F32)
        call("commu_gemm_tn_bf16_batched", _p(dsk), ld_dsk, TB * ld_dsk, _p(qv2), HD, DH, _p(slabs), DH, ld_dsk * DH, TB,
             ld_dsk, DH, ns, H, tri_B, tri_M, _s())
        call("commu_reduce_slabs2d_f32", _p(drd), drd.stride(0), DH, _p(slabs), K, DH, ns, ld_dsk * DH, H, 0, 1.0 / c2, _s())
    if defer is None:
        drd_part()
    else:
        defer(drd_part)
    _bias_grads(dq if dq_colsum else None, du_part, du, dvb, HD, dev, defer, colsum_group)
    return delta


def _bias_grads(dq, du_part, du, dvb, HD, dev, defer, group=None):
    """d r_w_bias += colsum(dq_ac) ; d r_r_bias += colsum(dq) - colsum(dq_ac)   (gradients only: deferrable).
    du_part holds the per-query-tile column sums of dq_ac; three column-sum launches, no temporaries (group: one slab pass
    now, the three final passes in the step's grouped launch)."""
    def bias_part():
        if defer is not None:          # local scratch outlives the caller on the deferring stream
            du_part.record_stream(torch.cuda.current_stream())
        colsum(du_part, du, group=group)
        if dq is not None:
            colsum(dq, dvb, group=group)
        colsum(du_part, dvb, alpha=-1.0, group=group)
    if defer is None:
        bias_part()
    else:
        defer(bias_part)


def sample_topk(logits, temperature, top_k, wrong=None, uniforms=None, active=None, token=None, probs_out=None,
                top_p=1.0):
    """In-place temperature + softmax + top-k + wrong-token mask + inverse-CDF draw per sequence.
    logits: fp32 [nseq, >=V] (row stride = ld), modified in place (logits[:, 1:V] /= temperature).
    top_p < 1: nucleus filter after the top-k / rejected-token step (an extra mode; the reference has top-k only)."""
    nseq = logits.shape[0]
    V = 729
    assert logits.dtype == F32 and logits.stride(1) == 1
    if token is None:
        token = torch.empty(nseq, device=logits.device, dtype=torch.int32)
    call("commu_sample_topk_topp", _p(logits), logits.stride(0), nseq, V, _p(wrong),
         0 if wrong is None else wrong.stride(0), _p(uniforms), _p(active), float(temperature), int(top_k), float(top_p),
         _p(token), _p(probs_out), 0 if probs_out is None else probs_out.stride(0), _s())
    return token


# ---------------------------------------------------------------------------------------------- fp32 parity mode
# (csrc/parity_f32.hip: the generation path on fp32 operands end to end -- the reference's arithmetic, train.py:48)
def _f32_2d(t, name):
    if not t.is_cuda:
        raise CommuHipError("commu_amd kernels need GPU tensors (no CPU fallback)")
    assert t.dtype == F32 and t.dim() == 2 and t.stride(1) == 1, f"{name}: fp32 row-major 2-D tensor expected"
    return t.stride(0)


def gemm_nt_f32(A, B, out=None, bias=None, resid=None, relu=False):
    """out fp32 [M, N] = A [M, K] . B [N, K]^T (+ bias) (ReLU) (+ resid): nn.Linear in fp32."""
    lda, ldb = _f32_2d(A, "A"), _f32_2d(B, "B")
    M, K = A.shape
    N = B.shape[0]
    assert B.shape[1] == K
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=F32)
    ldc = _f32_2d(out, "out")
    assert out.shape == (M, N)
    if bias is not None:
        assert bias.dtype == F32 and bias.is_contiguous() and bias.numel() >= N
    ldr = 0 if resid is None else _f32_2d(resid, "resid")
    call("commu_gemm_nt_f32", _p(A), lda, _p(B), ldb, _p(out), ldc, M, N, K, _p(bias), _p(resid), ldr, 1 if relu else 0, _s())
    return out


def embed_f32(tok, E, out=None):
    """model.py:409-420: E[tok] * sqrt(d_model), fp32."""
    if not tok.is_cuda:
        raise CommuHipError("commu_amd kernels need GPU tensors (no CPU fallback)")
    tok = tok.contiguous().view(-1)
    assert tok.dtype == torch.long and E.dtype == F32 and E.is_contiguous()
    rows, D = tok.numel(), E.shape[1]
    if out is None:
        out = torch.empty(rows, D, device=E.device, dtype=F32)
    call("commu_embed_f32", _p(tok), _p(E), _p(out), _f32_2d(out, "out"), rows, D, math.sqrt(D), _s())
    return out


def posemb_f32(inv_freq, n, D, out=None, clamp_len=-1):
    """model.py:142-147 by distance: row d = [sin(p * inv_freq) | cos(p * inv_freq)], p = d or min(d, clamp_len)."""
    if out is None:
        out = torch.empty(n, D, device=inv_freq.device, dtype=F32)
    assert inv_freq.dtype == F32 and inv_freq.numel() == D // 2
    call("commu_posemb_f32", _p(inv_freq), _p(out), _f32_2d(out, "out"), n, D, int(clamp_len), _s())
    return out


def layernorm_f32(x, gamma, beta, eps=1e-5, out=None):
    ldx = _f32_2d(x, "x")
    rows, D = x.shape
    if out is None:
        out = torch.empty(rows, D, device=x.device, dtype=F32)
    assert gamma.dtype == F32 and beta.dtype == F32 and gamma.numel() == D
    call("commu_layernorm_f32", _p(x), ldx, _p(gamma), _p(beta), _p(out), _f32_2d(out, "out"), rows, D, float(eps), _s())
    return out


def relattn_f32(q, k, v, stride_key, stride_seq, rd, u, vb, T, M, B, H, DH, same_length, mem_len, scale, klen=None,
                reset=None, out=None):
    """model.py:283-345 in fp32 (see include/commu_hip.h, commu_relattn_f32).  q: 2-D view [T*B, >= H*DH]; k, v: tensors whose
    data pointer is element (key 0, sequence 0, head 0, 0); rd [>= max distance + 1, >= H*DH]."""
    ld_q = _f32_2d(q, "q")
    ld_rd = _f32_2d(rd, "rd")
    assert k.dtype == F32 and v.dtype == F32 and u.dtype == F32 and vb.dtype == F32
    if out is None:
        out = torch.empty(T * B, H * DH, device=q.device, dtype=F32)
    call("commu_relattn_f32", _p(q), ld_q, _p(k), _p(v), int(stride_key), int(stride_seq), _p(rd), ld_rd, _p(u), _p(vb),
         _p(klen), _p(reset), _p(out), _f32_2d(out, "out"), T, M, B, H, DH, 1 if same_length else 0, int(mem_len), float(scale),
         _s())
    return out


def decode_kv_append_f32(qkv, kc, vc, klen, active, HD, Lmax):
    call("commu_decode_kv_append_f32", _p(qkv), _f32_2d(qkv, "qkv"), _p(kc), _p(vc), _p(klen), _p(active), qkv.shape[0], HD,
         Lmax, _s())
