"""ctypes binding of libphenaki_hip.so (include/phenaki_hip.h).

The product has NO CPU fallback: importing this module is harmless, but the first kernel call without the
built library (python -m phenaki_pytorch_amd.build) or without a HIP device raises.
PyTorch is used only as the owner of device memory and of the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PK_LIB_PATH: load another build of the same C ABI (A/B timing of two builds on one GPU box; never a fallback)
LIB_PATH = os.environ.get('PK_LIB_PATH') or os.path.join(_HERE, 'libphenaki_hip.so')

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_long
_LL = ctypes.c_longlong
_ULL = ctypes.c_ulonglong

# name -> argtypes, kept in the order of include/phenaki_hip.h (tests check every symbol is exported)
SIGNATURES = {
    'pk_gemm': [_I, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P, _P],
    'pk_gemm_ex': [_I, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P, _I, _I, _P, _I, _P, _P, _F, _P, _P, _P, _P, _I, _P],
    'pk_gemm_auto_variant': [_I, _I, _I, _I, _I, _I, _I, _I],
    'pk_layernorm': [_P, _I, _P, _P, _F, _P, _I, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'pk_rmsnorm': [_P, _I, _P, _F, _P, _P, _I, _I, _I, _I, _P],
    'pk_gated_gelu_tanh': [_P, _I, _P, _I, _I, _I, _I, _P],
    'pk_l2norm_rows': [_P, _I, _P, _I, _I, _I, _I, _P],
    'pk_patchify_ln': [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _F, _P, _I, _I, _P],
    'pk_patch_embed': [_P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I,
                       _P, _I, _P, _P, _P, _I, _I, _I,
                       _P, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    'pk_patch_embed_slices': [_I],
    'pk_patch_embed_splitk': [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                              _P, _I, _P, _P, _I, _I, _I,
                              _P, _I, _P, _P, _I, _I, _I, _P],
    'pk_patch_embed_finish': [_P, _P, _I, _I, _I, _I, _P, _P, _F, _P, _P, _F, _P, _I, _P, _I, _I, _I, _I, _P],
    'pk_patch_embed_finish_groups': [_P, _I, _I, _P, _I, _P, _I, _P],
    'pk_patch_frame_mask': [_P, _LL, _P, _LL, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'pk_unpatchify': [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'pk_sqdiff_partials': [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    'pk_peg': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'pk_lfq_encode': [_P, _I, _P, _P, _P, _P, _I, _I, _I, _P],
    'pk_lfq_decode': [_P, _P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _P],
    'pk_lfq_aux_parts': [_I],
    'pk_lfq_aux_prep': [_P, _I, _I, _F, _F, _P, _P, _P, _P, _P],
    'pk_lfq_aux_codebook': [_P, _I, _F, _F, _P, _P, _P],
    'pk_lfq_aux_grad': [_P, _P, _P, _I, _I, _F, _F, _F, _F, _P, _P],
    'pk_lfq_aux_finish': [_P, _P, _I, _P, _I, _F, _F, _F, _P, _P],
    'pk_layernorm_lfq': [_P, _I, _P, _P, _F, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    'pk_embed': [_P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _P],
    'pk_cpb_input': [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    'pk_attn_pads': [_I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)],
    'pk_attn_prep': [_I, _P, _I, _P, _I, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'pk_qkv_project': [_I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _I, _I, _P, _P],
    'pk_qkv_attn': [_I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _L, _I, _P, _I, _P, _I, _P, _P],
    'pk_q_attn_cached': [_I, _P, _I, _P, _I, _I, _I, _I, _I, _P, _F, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P],
    'pk_attn_fwd': [_I, _P, _P, _P, _P, _L, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _I, _F, _P],
    'pk_attn_fwd_lse': [_I, _P, _P, _P, _P, _L, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    'pk_attn_small': [_P, _I, _P, _I, _P, _P, _F, _P, _L, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    'pk_cfg_mix': [_P, _I, _I, _I, _I, _P, _I, _F, _I, _P, _I, _I, _I, _P],
    'pk_vocab_ntiles': [_I],
    'pk_vocab_sample_philox': [_I, _P, _I, _P, _I, _P, _I, _I, _I, _F, _P, _ULL, _ULL, ctypes.c_uint, _I, _P, _P],
    'pk_vocab_sample': [_I, _P, _I, _P, _I, _P, _I, _I, _I, _F, _P, _P, _ULL, _P, _I, _P, _P],
    'pk_vocab_reduce': [_P, _I, _I, _P, _P, _P, _P, _P, _I, _P],
    'pk_vocab_ce': [_I, _P, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P],
    'pk_ce_grad_slab': [_I, _P, _I, _P, _P, _P, _I, _I, _I, _F, _P, _P, _I, _P, _I, _P, _P],
    'pk_topk_mask': [_P, _I, _I, _I, _LL, _P, _P, _P, _P, _P],
    'pk_critic_head': [_P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P, _F, _ULL, _P, _P, _P],
    'pk_pack': [_P, _LL, _P, _I, _I, _I, _P, _LL, _I, _I, _P],
    'pk_scatter_rows': [_P, _LL, _P, _P, _LL, _I, _I, _P],
    'pk_colsum_parts': [_I],
    'pk_colsum': [_P, _LL, _I, _I, _F, _P, _I, _P, _P],
    'pk_ln_bwd_parts': [_I],
    'pk_layernorm_bwd': [_P, _LL, _P, _P, _LL, _P, _LL, _P, _LL, _P, _P, _F, _I, _I, _P],
    'pk_geglu': [_P, _LL, _I, _P, _LL, _I, _I, _P],
    'pk_geglu_bwd': [_P, _LL, _I, _P, _LL, _P, _LL, _I, _I, _P],
    'pk_leaky_bwd': [_P, _LL, _P, _LL, _P, _LL, _I, _I, _F, _P],
    'pk_scaled_diff': [_P, _P, _F, _P, _P, _LL, _P],
    'pk_sign': [_P, _F, _P, _LL, _P],
    'pk_mul': [_P, _P, _P, _LL, _P],
    'pk_peg_wgrad_parts': [_LL],
    'pk_peg_bwd': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'pk_embed_bwd': [_P, _P, _F, _P, _P, _I, _I, _I, _P],
    'pk_bias_gather': [_P, _I, _P, _I, _P, _I, _I, _P],
    'pk_bias_scatter': [_P, _P, _I, _P, _I, _I, _I, _P],
    'pk_sum_batch': [_P, _LL, _I, _P, _LL, _P],
    'pk_sum_batch_multi': [_P, _I, _P],
    'pk_peg_adjoint': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'pk_colsum_multi': [_P, _I, _P],
    'pk_reduce_multi': [_P, _I, _P, _I, _P],
    'pk_pack_multi': [_P, _I, _P],
    'pk_pack_table_prepare': [_P, _I],
    'pk_pack_table': [_P, _I, _I, _P],
    'pk_bce_head': [_P, _LL, _P, _P, _P, _F, _P, _P, _P, _P, _LL, _P, _P, _I, _I, _P],
    'pk_attn_train_prep': [_P, _LL, _P, _LL, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'pk_attn_train_prep_bwd': [_P, _LL, _P, _LL, _P, _P, _P, _F, _P, _P, _P, _P, _LL, _P, _LL, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'pk_gemm_splitk': [_I, _P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P],
    'pk_adamw': [_P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _LL, _P],
    'pk_adamw_multi': [_P, _I, _F, _F, _F, _F, _F, _I, _P],
    'pk_attn_bwd': [_P, _P, _P, _P, _LL, _I, _P, _LL, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'pk_attn_bwd_ws': [_P, _P, _P, _P, _LL, _I, _P, _LL, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _LL, _P],
    'pk_attn_bwd_work': [_I, _I, _I, _I, _I],
    'pk_im2col': [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _LL, _P],
    'pk_col2im': [_P, _LL, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    'pk_nchw_to_rows': [_P, _I, _I, _I, _I, _I, _P, _P],
    'pk_rows_to_nchw': [_P, _I, _I, _I, _I, _I, _P, _P],
    'pk_pick_frames': [_P, _P, _I, _I, _I, _I, _I, _P, _I, _P],
    'pk_bmm': [_P, _LL, _LL, _I, _P, _LL, _LL, _I, _P, _LL, _LL, _I, _I, _I, _I, _I, _P],
    'pk_row_softmax': [_P, _P, _P, _P, _LL, _I, _I, _P],
    'pk_row_l2scale': [_P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _P],
    'pk_row_ln_bwd2': [_P, _P, _P, _P, _P, _F, _P, _P, _P, _LL, _I, _P],
}

_ERR = {-1: 'PK_EINVAL (bad shape/size/flag)', -2: 'PK_EALIGN (pointer/stride alignment)', -3: 'PK_ELAUNCH (HIP launch failed)'}

_lib = None


def load():
    """dlopen the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: run `python -m phenaki_pytorch_amd.build` '
                               '(the MI355X build has no CPU / eager fallback)')
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _I
        _lib = lib
    return _lib


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f'{name} failed: {_ERR.get(rc, rc)}')


def ptr(t):
    return None if t is None else t.data_ptr()


def f32p(t, name='parameter', rows_ok=False):
    """device pointer of an f32 parameter / buffer the kernels read as `const float*`: a module that went through
    .half() / .bfloat16() / .double(), or holds a strided view, must fail here instead of computing garbage."""
    if t is None:
        return None
    if t.dtype != torch.float32 or not (t.is_contiguous() or (rows_ok and t.stride(-1) == 1)):
        raise RuntimeError(f'{name}: the MI355X kernels read this tensor as contiguous float32, got {t.dtype} '
                           f'{"contiguous" if t.is_contiguous() else "strided"} (keep the module in float32; the compute '
                           'precision is chosen with set_compute_dtype, not with .half()/.bfloat16())')
    return t.data_ptr()


def stream(t=None):
    """the HIP stream of the operand's device.  Kernels are launched on the CURRENT device, so an operand living on another
    GPU of the process is refused (torch.cuda.set_device / one process per GPU) instead of being launched on the wrong stream."""
    if t is None:
        return torch.cuda.current_stream().cuda_stream
    idx = t.device.index
    cur = _current_device()
    if idx is not None and idx != cur:
        raise RuntimeError(f'operand on cuda:{idx} but the current device is cuda:{cur}: '
                           'call torch.cuda.set_device(...) first (one process per GPU)')
    # the raw handle straight from the C binding: torch.cuda.current_stream() builds a Stream object per call (5.7 us -- more than
    # a third of the host cost of a launch when the sampling loop is launch-bound at small batches)
    return _raw_stream(cur)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or (lambda i: torch.cuda.current_stream(i).cuda_stream)
_current_device = getattr(torch._C, '_cuda_getDevice', None) or torch.cuda.current_device


def require_device(t, name='tensor'):
    if not t.is_cuda:
        raise RuntimeError(f'{name} is on {t.device}: the MI355X build runs on HIP devices only (no CPU fallback)')


# compute dtypes of the C ABI: exact f32 MFMA | bf16 operands | split-bf16 ("bf16x3": every GEMM / attention product as
# hi.hi + hi.lo + lo.hi on the bf16 matrix cores, operands x = bf16(x) + bf16(x - bf16(x)); activations stay f32 in memory)
F32, BF16, BF16X3 = 0, 1, 2
ACT_NONE, ACT_GEGLU, ACT_LEAKY = 0, 1, 2


def tdtype(dtype):
    """torch dtype of the activations a GEMM of this compute dtype reads / writes as `T`"""
    return torch.bfloat16 if dtype == BF16 else torch.float32


def split_planes(w32):
    """(N, K) f32, K % 32 == 0 -> the pre-split image the bf16x3 kernels read as W (csrc/common.hpp): per row and per block of 32
    k-elements 64 bf16 = [hi x 32 | lo x 32], hi = bf16(w), lo = bf16(w - hi); returned as an (N, K) float32-TYPED tensor of raw bits
    (same bytes per row as f32, so strides / the LDS-DMA ring are those of the f32 layout)."""
    n, k = w32.shape
    assert k % 32 == 0 and w32.dtype == torch.float32
    hi = w32.to(torch.bfloat16)
    lo = (w32 - hi.float()).to(torch.bfloat16)
    img = torch.cat((hi.view(n, k // 32, 32), lo.view(n, k // 32, 32)), dim=-1).contiguous()       # (n, k/32, 64) bf16
    return img.view(n, 2 * k).view(torch.float32)


# ----------------------------------------------------------------------------- wrappers

def gemm(dtype, A, W, M, N, K, *, C, bias=None, res=None, act=ACT_NONE, a_rows=None, lda=None, ldc=None, variant=0, C2=None, ln=None,
         scatter=None, stats_out=None, ln_stats=None, dup_rows=0):
    """C = act(A @ W^T + bias) (+ res).  A: (rows, K) f32 or T; W: (N, Kpad) T; C preallocated.
    a_rows gathers rows of A (A.shape[0] physical rows bound the DMA descriptor).
    C2: optional bf16 copy of an f32 C.  ln = (s, t, eps): the LayerNorm in front of this Linear folded in (A = the un-normalised
    rows, W = gamma (.) W, s / t its (N,) correction vectors).  scatter = (row_off (M,) int32, col_off (N,) int32): element (m, n) is
    written to C.view(-1)[row_off[m] + col_off[n]] (f32 C of any shape; the un-patchify map).  stats_out (M, ceil(N/32), 2) f32: per-chunk
    (sum, sum of squares) of every output row for the LayerNorm-folded GEMM that consumes it, which passes the same buffer as ln_stats.
    dup_rows: every output row m of C / C2 is also written at row m + dup_rows."""
    a_is_f32 = 1 if A.dtype == torch.float32 else 0
    out_is_f32 = 1 if C.dtype == torch.float32 else 0
    lda = A.stride(-2) if lda is None else lda
    ldc = (C.stride(-2) if scatter is None else N) if ldc is None else ldc
    ldr = res.stride(-2) if res is not None else 0
    rc = load().pk_gemm_ex(dtype, a_is_f32, ptr(A), lda, ptr(W), W.stride(0), M, N, K, f32p(bias, 'bias'), f32p(res, 'residual'), ldr,
                           ptr(C), ldc, out_is_f32, act, ptr(a_rows), A.shape[0] if a_rows is not None else M, variant,
                           ptr(C2), C2.stride(-2) if C2 is not None else 0, ptr(ln[0]) if ln else None, ptr(ln[1]) if ln else None,
                           float(ln[2]) if ln else 0., ptr(scatter[0]) if scatter else None, ptr(scatter[1]) if scatter else None,
                           ptr(stats_out) if stats_out is not None else None, ptr(ln_stats) if ln_stats is not None else None, int(dup_rows), stream(C))
    _check(rc, 'pk_gemm_ex')
    return C


def layernorm(x, gamma, beta, M, D, *, out=None, out2=None, raw=None, eps=1e-5, remap=(0, 0, 0), perm=(0, 0), ldx=None):
    """out (T) / out2 (f32) <- LN(x); raw (T, same type as out) <- x.  remap / perm: output row maps (see the header)."""
    tt = out if out is not None else raw
    out_kind = 1 if (tt is not None and tt.dtype == torch.bfloat16) else 0
    rc = load().pk_layernorm(f32p(x, 'x', rows_ok=True), x.stride(-2) if ldx is None else ldx, f32p(gamma, 'LayerNorm gamma'), f32p(beta, 'LayerNorm beta'), eps,
                             ptr(out), out.stride(-2) if out is not None else 0, out_kind,
                             ptr(out2), out2.stride(-2) if out2 is not None else 0,
                             ptr(raw), raw.stride(-2) if raw is not None else 0, M, D, *remap, *perm, stream(x))
    _check(rc, 'pk_layernorm')


def patchify_ln(video, f0, nt, pt, ph, pw, weight, bias, out, eps=1e-5):
    B, C, F, H, W = video.shape
    rc = load().pk_patchify_ln(f32p(video, 'video'), B, C, F, H, W, f0, nt, pt, ph, pw, f32p(weight, 'LayerNorm weight'), f32p(bias, 'LayerNorm bias'), eps,   # weight = bias = None: raw rows
                               ptr(out), out.stride(0), 1 if out.dtype == torch.bfloat16 else 0, stream(video))
    _check(rc, 'pk_patchify_ln')


def patch_embed(video, ph, pw, N, groups, eps=1e-5):
    """groups: 1 or 2 tuples (Wg (N, Kpad) bf16 = gamma (.) W, s (N,), t (N,), out (rows, N) f32, f0, nt, pt), long-K group first.
    out_g = LayerNorm_P(patches of frames [f0, f0 + nt*pt)) @ W^T + b, the patch rows gathered from the video inside the kernel."""
    B, C, F, H, W = video.shape
    flat = []
    for (Wg, s, t, out, f0, nt, pt) in groups:
        flat += [ptr(Wg), Wg.stride(0), f32p(s, 'folded s'), f32p(t, 'folded t'), ptr(out), f0, nt, pt]
    if len(groups) == 1:
        flat += [None, 0, None, None, None, 0, 0, 0]
    ldo = groups[0][3].stride(0)
    rc = load().pk_patch_embed(f32p(video, 'video'), B, C, F, H, W, ph, pw, N, eps, len(groups), *flat, ldo, stream(video))
    _check(rc, 'pk_patch_embed')


def patch_embed_slices(K):
    return load().pk_patch_embed_slices(int(K))


def patch_embed_splitk(video, ph, pw, N, groups):
    """row panels x all columns x K-slices (the header): groups = 1 or 2 tuples (Wg (N, Kpad) bf16, part (slices, rows, N) f32,
    stats (slices, rows, 2) f32, f0, nt, pt), long-K group first; finish every group with patch_embed_finish"""
    B, C, F, H, W = video.shape
    flat = []
    for (Wg, part, stats, f0, nt, pt) in groups:
        flat += [ptr(Wg), Wg.stride(0), ptr(part), ptr(stats), f0, nt, pt]
    if len(groups) == 1:
        flat += [None, 0, None, None, 0, 0, 0]
    rc = load().pk_patch_embed_splitk(f32p(video, 'video'), B, C, F, H, W, ph, pw, N, len(groups), *flat, stream(video))
    _check(rc, 'pk_patch_embed_splitk')


def patch_embed_finish(part, stats, K, s, t, eps1, gamma2, beta2, eps2, *, out2=None, out=None, remap=(0, 0, 0)):
    """tokens of one frame group from its K-slices: folded LayerNorm(P) + bias, then LayerNorm(N); out2 f32 / out bf16 rows (remapped)"""
    nslices, rows, N = part.shape
    rc = load().pk_patch_embed_finish(ptr(part), ptr(stats), nslices, rows, N, int(K), f32p(s, 'folded s'), f32p(t, 'folded t'), float(eps1),
                                      f32p(gamma2, 'LayerNorm weight'), f32p(beta2, 'LayerNorm bias'), float(eps2),
                                      ptr(out2), out2.stride(-2) if out2 is not None else 0, ptr(out), out.stride(-2) if out is not None else 0,
                                      *remap, stream(part))
    _check(rc, 'pk_patch_embed_finish')


class _PatchFinishGroup(ctypes.Structure):
    _fields_ = [('part', _P), ('stats', _P), ('nslices', _I), ('rows', _I), ('K', _I), ('s', _P), ('t', _P), ('eps1', _F), ('gamma2', _P), ('beta2', _P),
                ('eps2', _F), ('remap_in', _I), ('remap_out', _I), ('remap_off', _I)]


def patch_embed_finish_groups(groups, *, out2=None, out=None):
    """pk_patch_embed_finish for 1 or 2 frame groups in ONE launch; groups: tuples (part, stats, K, s, t, eps1, gamma2, beta2, eps2, remap)"""
    arr = (_PatchFinishGroup * len(groups))()
    for d, (part, stats, K, s, t, eps1, gamma2, beta2, eps2, remap) in zip(arr, groups):
        nslices, rows, N = part.shape
        d.part, d.stats, d.nslices, d.rows, d.K = ptr(part), ptr(stats), nslices, rows, int(K)
        d.s, d.t, d.eps1 = f32p(s, 'folded s'), f32p(t, 'folded t'), float(eps1)
        d.gamma2, d.beta2, d.eps2 = f32p(gamma2, 'LayerNorm weight'), f32p(beta2, 'LayerNorm bias'), float(eps2)
        d.remap_in, d.remap_out, d.remap_off = remap
    rc = load().pk_patch_embed_finish_groups(ctypes.cast(arr, _P), len(groups), N, ptr(out2), out2.stride(-2) if out2 is not None else 0,
                                             ptr(out), out.stride(-2) if out is not None else 0, stream(groups[0][0]))
    _check(rc, 'pk_patch_embed_finish_groups')


def patch_frame_mask(src, dst, fmask, video_shape, f0, nt, pt, ph, pw):
    """dst = src with the patch-layout elements of the frames `fmask` (B, F) uint8 drops set to zero"""
    B, C, F, H, W = video_shape
    rc = load().pk_patch_frame_mask(ptr(src), src.stride(0), ptr(dst), dst.stride(0), ptr(fmask), B, C, F, H, W, f0, nt, pt, ph, pw, stream(src))
    _check(rc, 'pk_patch_frame_mask')
    return dst


def unpatchify(pix, video, f0, nt, pt, ph, pw):
    B, C, F, H, W = video.shape
    rc = load().pk_unpatchify(ptr(pix), pix.stride(0), ptr(video), B, C, F, H, W, f0, nt, pt, ph, pw, stream(video))
    _check(rc, 'pk_unpatchify')


def sqdiff_sum(a, b, frame_mask=None):
    """sum((a - b)^2) over the frames `frame_mask` (B, F) keeps (all if None) of two (B, C, F, H, W) f32 videos -> 0-d f64."""
    B, C, F, H, W = a.shape
    lib = load()
    partials = torch.empty((1024,), device=a.device, dtype=torch.float64)            # PK_SQDIFF_BLOCKS
    fm = frame_mask.to(torch.uint8).contiguous() if frame_mask is not None else None
    _check(lib.pk_sqdiff_partials(ptr(a), ptr(b), ptr(fm), B, C, F, H, W, ptr(partials), stream(a)), 'pk_sqdiff_partials')
    return partials.sum()


def peg(x, wt, bias, out, B, T, H, W, D, causal, out_t=None):
    rc = load().pk_peg(f32p(x, 'x', rows_ok=True), f32p(wt, 'PEG weight'), f32p(bias, 'PEG bias'), ptr(out), ptr(out_t), B, T, H, W, D,
                       1 if causal else 0, stream(x))
    _check(rc, 'pk_peg')


def lfq_encode(x, wp, bp, ids, proj, M, D, cd):
    rc = load().pk_lfq_encode(ptr(x), x.stride(-2), f32p(wp, 'LFQ project_in.weight'), f32p(bp, 'LFQ project_in.bias'), ptr(ids), ptr(proj), M, D, cd, stream(x))
    _check(rc, 'pk_lfq_encode')


def lfq_decode(ids, wo, bo, out, M, D, cd, *, ids_prime=None, perm=(0, 0)):
    """out[orow] = project_out(+-1 bits of id[row]); ids_prime (nb, n_prime) int64: the primed tokens in front of every sequence of
    ids (nb, n) (M = nb * (n_prime + n)); perm = (pb, pc): rows (a, b, c) are written at (a, c, b)"""
    n_prime = ids_prime.shape[-1] if ids_prime is not None else 0
    n = ids.shape[-1] if ids_prime is not None else 0
    rc = load().pk_lfq_decode(ptr(ids), f32p(wo, 'LFQ project_out.weight'), f32p(bo, 'LFQ project_out.bias'), ptr(out), M, D, cd,
                              ptr(ids_prime), n_prime, n, perm[0], perm[1], stream(out))
    _check(rc, 'pk_lfq_decode')


def lfq_aux(proj, *, inv_temperature=100., codebook_scale=1., entropy_loss_weight=0.1, commitment_loss_weight=0.25, diversity_gamma=1.):
    """training-mode auxiliary loss of the LFQ for proj (M, cd) f32 = project_in(x) (the header's pk_lfq_aux_* group):
    returns (out (4,) f32 = [aux, per-sample entropy, codebook entropy, commitment], dproj (M, cd) = d aux / d proj)."""
    lib = load()
    M, cd = proj.shape
    dev = proj.device
    hi, lo = (cd + 1) // 2, cd // 2
    NA, NB = 1 << hi, 1 << lo
    f32 = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
    A, B, ent, commit = f32(M, NA), f32(M, NB), f32(M), f32(M)
    alpha = 4. * float(inv_temperature) * float(codebook_scale)
    st = stream(proj)
    _check(lib.pk_lfq_aux_prep(f32p(proj, 'LFQ projection'), M, cd, alpha, float(codebook_scale), ptr(A), ptr(B), ptr(ent), ptr(commit), st), 'pk_lfq_aux_prep')
    Q = bmm(A, B, f32(NA, NB), True, False, 1, NA, NB, M, lda=NA, ldb=NB, ldc=NB)                       # A^T B = M x the batch distribution
    G, hc = f32(NA, NB), f32(lib.pk_lfq_aux_parts(cd))
    w_e, w_c, gamma = float(entropy_loss_weight), float(commitment_loss_weight), float(diversity_gamma)
    _check(lib.pk_lfq_aux_codebook(ptr(Q), cd, 1. / M, -w_e * gamma / M, ptr(G), ptr(hc), st), 'pk_lfq_aux_codebook')
    GA = bmm(B, G, f32(M, NA), False, True, 1, M, NA, NB, lda=NB, ldb=NB, ldc=NA)                      # B G^T
    GB = bmm(A, G, f32(M, NB), False, False, 1, M, NB, NA, lda=NA, ldb=NB, ldc=NB)                     # A G
    dproj, out = f32(M, cd), f32(4)
    _check(lib.pk_lfq_aux_grad(ptr(proj), ptr(GA), ptr(GB), M, cd, alpha, float(codebook_scale), w_e / M, 2. * w_c / (M * cd), ptr(dproj), st), 'pk_lfq_aux_grad')
    _check(lib.pk_lfq_aux_finish(ptr(ent), ptr(commit), M, ptr(hc), cd, w_e, gamma, w_c, ptr(out), st), 'pk_lfq_aux_finish')
    return out, dproj


def embed(ids, tok, pos, out, S, n, D, *, nb=None, ids_prime=None, out_t=None):
    """out[s*n_tot + i] = tok[id] + pos[i]; ids (nb, n) int64 shared by the S sequences (s % nb), ids_prime (nb, n_prime) optional"""
    n_prime = ids_prime.shape[-1] if ids_prime is not None else 0
    rc = load().pk_embed(ptr(ids_prime), n_prime, ptr(ids), n, S if nb is None else nb, f32p(tok, 'token_emb.weight'),
                         f32p(pos, 'pos_emb.weight'), ptr(out), ptr(out_t), S, D, stream(out))
    _check(rc, 'pk_embed')


def layernorm_lfq(x, gamma, beta, wp, bp, ids, M, D, cd, *, tokens=None, proj=None, eps=1e-5, perm=(0, 0)):
    """ids[orow] = LFQ(LayerNorm(x[row])): the encoder's final norm_out fused with the quantizer (cd <= 16)"""
    rc = load().pk_layernorm_lfq(f32p(x, 'x', rows_ok=True), x.stride(-2), f32p(gamma, 'LayerNorm gamma'), f32p(beta, 'LayerNorm beta'), eps,
                                 f32p(wp, 'LFQ project_in.weight'), f32p(bp, 'LFQ project_in.bias'), ptr(ids), ptr(tokens),
                                 tokens.stride(-2) if tokens is not None else 0, ptr(proj), M, D, cd, *perm, stream(x))
    _check(rc, 'pk_layernorm_lfq')


def cpb_input(w0, b0, out, dims, D):
    nd = len(dims)
    d = (1,) * (3 - nd) + tuple(dims)
    rc = load().pk_cpb_input(ptr(w0), ptr(b0), ptr(out), d[0], d[1], d[2], nd, D, stream(out))
    _check(rc, 'pk_cpb_input')


def attn_pads(nq, n_kv, nnull):
    a, b = _I(0), _I(0)
    _check(load().pk_attn_pads(nq, n_kv, nnull, ctypes.byref(a), ctypes.byref(b)), 'pk_attn_pads')
    return a.value, b.value


def attn_prep(dtype, q, kv, null_kv, q_scale, k_scale, scale, Qp, Kp, Vt, S, h, nq, n_kv, nnull):
    """q_scale = k_scale = None: plain dot-product attention operands (no l2norm; q * scale) -- the T5 text encoder"""
    rc = load().pk_attn_prep(dtype, ptr(q), q.stride(-2), ptr(kv), kv.stride(-2) if kv is not None else 0, f32p(null_kv, 'null_kv') if nnull else None,
                             f32p(q_scale, 'q_scale'), f32p(k_scale, 'k_scale'), scale, ptr(Qp), ptr(Kp), ptr(Vt), S, h, nq, n_kv, nnull, stream(q))
    _check(rc, 'pk_attn_prep')


def rmsnorm(x, w, M, D, out, eps=1e-6, rowmask=None):
    rc = load().pk_rmsnorm(f32p(x, 'x', rows_ok=True), x.stride(-2), f32p(w, 'T5LayerNorm weight'), eps, ptr(rowmask), ptr(out), out.stride(-2),
                           1 if out.dtype == torch.bfloat16 else 0, M, D, stream(x))
    _check(rc, 'pk_rmsnorm')


def gated_gelu_tanh(h, out, M, F):
    rc = load().pk_gated_gelu_tanh(ptr(h), h.stride(-2), ptr(out), out.stride(-2), 1 if out.dtype == torch.bfloat16 else 0, M, F, stream(h))
    _check(rc, 'pk_gated_gelu_tanh')


def _qkv_dtype(xq):
    """operand type of the fused projection kernels from the rows they read: bf16 rows -> 1 (bf16), f32 rows -> 2 (split-bf16)"""
    if xq.dtype == torch.bfloat16:
        return BF16
    if xq.dtype == torch.float32:
        return BF16X3
    raise RuntimeError(f'fused projection kernels take bf16 (bf16 mode) or f32 (bf16x3 mode) rows, got {xq.dtype}')


def qkv_project(xq, xkv, wq, wkv, S, nseq, h, K, q_scale, k_scale, scale, Qp, Kp, Vt, nq_pad, nk_pad, q_ln_s=None):
    rc = load().pk_qkv_project(_qkv_dtype(xq), ptr(xq), ptr(xkv), xq.stride(-2), ptr(wq), ptr(wkv), wq.stride(0), S, nseq, h, K, f32p(q_scale, 'q_scale'),
                                f32p(k_scale, 'k_scale'), scale, ptr(Qp), ptr(Kp), ptr(Vt), nq_pad, nk_pad, ptr(q_ln_s), stream(xq))
    _check(rc, 'pk_qkv_project')


def qkv_attn(xq, xkv, wq, wkv, S, n, h, K, q_scale, k_scale, scale, O, *, bias=None, slopes=None, causal=False, q_ln_s=None):
    """O (S*n, h*64) <- softmax(l2norm(xq Wq^T) l2norm(xkv Wk^T)^T * scale + bias) (xkv Wv^T), n <= 64 (one launch); bf16 rows / bf16 O, or
    f32 rows / f32 O with split-bf16 products"""
    bh, bld = (bias.stride(0), bias.stride(1)) if bias is not None else (0, 0)
    rc = load().pk_qkv_attn(_qkv_dtype(xq), ptr(xq), ptr(xkv), xq.stride(-2), ptr(wq), ptr(wkv), wq.stride(0), S, n, h, K, f32p(q_scale, 'q_scale'),
                            f32p(k_scale, 'k_scale'), scale, f32p(bias, 'attention bias'), bh, bld, f32p(slopes, 'ALiBi slopes'),
                            1 if causal else 0, ptr(O), O.stride(-2), ptr(q_ln_s), stream(xq))
    _check(rc, 'pk_qkv_attn')


def q_attn_cached(xq, wq, S, n, h, K, q_scale, scale, Kp, Vt, nk_pad, n_kv, nnull, O, *, kmask=None, q_ln_s=None):
    """O (S*n, h*64) <- cross-attention of the rows of xq against the cached K^ / V^T images (one launch; n % 64 == 0, <= 64 keys)"""
    rc = load().pk_q_attn_cached(_qkv_dtype(xq), ptr(xq), xq.stride(-2), ptr(wq), wq.stride(0), S, n, h, K, f32p(q_scale, 'q_scale'), scale, ptr(q_ln_s),
                                 ptr(Kp), ptr(Vt), nk_pad, n_kv, nnull, ptr(kmask), ptr(O), O.stride(-2), stream(xq))
    _check(rc, 'pk_q_attn_cached')


def attn_fwd(dtype, Qp, Kp, Vt, O, S, h, nq, n_kv, nnull, *, bias=None, kmask=None, slopes=None, causal=False, bias_table=None,
             score_bound=None, lse=None):
    """bias: full (h, nq, n_kv) f32 tensor, or bias_table = (tab (h, L) f32, pos_code (n,) int32, offset, ...): the relative-position
    form.  score_bound: upper bound of sim + bias (python float) -> fixed-offset softmax; None: running-max flash loop.
    lse ((S h nq,) f32; the training forward): also write every row's log-sum-exp for pk_attn_bwd."""
    tab, codes, off = bias_table[:3] if bias_table is not None else (None, None, 0)
    run4 = 1 if (bias_table is not None and len(bias_table) > 5 and bias_table[5]) else 0
    if bias is not None:
        bh, bld = bias.stride(0), bias.stride(1)
    else:
        bh, bld = 0, 0
    if lse is not None:
        assert bias_table is None and score_bound is None
        rc = load().pk_attn_fwd_lse(dtype, ptr(Qp), ptr(Kp), ptr(Vt), ptr(bias), bh, bld, ptr(kmask), f32p(slopes, 'ALiBi slopes'),
                                    1 if causal else 0, ptr(O), O.stride(-2), 1 if O.dtype == torch.float32 else 0, S, h, nq, n_kv, nnull, ptr(lse), stream(O))
        _check(rc, 'pk_attn_fwd_lse')
        return
    rc = load().pk_attn_fwd(dtype, ptr(Qp), ptr(Kp), ptr(Vt), ptr(bias), bh, bld, ptr(kmask), f32p(slopes, 'ALiBi slopes'),
                            1 if causal else 0, ptr(O), O.stride(-2), 1 if O.dtype == torch.float32 else 0,
                            S, h, nq, n_kv, nnull, f32p(tab, 'bias table'), tab.shape[1] if tab is not None else 0, ptr(codes), off, run4,
                            float('nan') if score_bound is None else float(score_bound), stream(O))
    _check(rc, 'pk_attn_fwd')


def attn_small(q, kv, q_scale, k_scale, scale, O, S, h, n, *, bias=None, kmask=None, slopes=None, causal=False):
    bh, bld = (bias.stride(0), bias.stride(1)) if bias is not None else (0, 0)
    rc = load().pk_attn_small(ptr(q), q.stride(-2), ptr(kv), kv.stride(-2), f32p(q_scale, 'q_scale'), f32p(k_scale, 'k_scale'), scale, ptr(bias), bh, bld,
                              ptr(kmask), f32p(slopes, 'ALiBi slopes'), 1 if causal else 0, ptr(O), O.stride(-2),
                              1 if O.dtype == torch.bfloat16 else 0, S, h, n, stream(q))
    _check(rc, 'pk_attn_small')


def cfg_mix(x, nb, n_tot, n_prime, rows, nrows, scale, has_null, out, D):
    rc = load().pk_cfg_mix(ptr(x), x.stride(-2), nb, n_tot, n_prime, ptr(rows), nrows, scale, 1 if has_null else 0,
                           ptr(out), out.stride(-2), 1 if out.dtype == torch.float32 else 0, D, stream(x))
    _check(rc, 'pk_cfg_mix')


def vocab_ntiles(V):
    return load().pk_vocab_ntiles(V)


def l2norm_rows(x, out, M, D):
    rc = load().pk_l2norm_rows(ptr(x), x.stride(-2), ptr(out), out.stride(-2), 1 if out.dtype == torch.bfloat16 else 0, M, D, stream(x))
    _check(rc, 'pk_l2norm_rows')


def vocab_sample(dtype, A, W, bias, M, V, D, temperature, U, rows, seed, need_lse, partials, no_noise=False, seed_dev=None):
    rc = load().pk_vocab_sample(dtype, ptr(A), A.stride(-2), ptr(W), W.stride(0), f32p(bias, 'to_logits.bias'), M, V, D, temperature,
                                ptr(U), ptr(rows), seed & 0xFFFFFFFFFFFFFFFF, ptr(seed_dev), (1 if need_lse else 0) | (2 if no_noise else 0),
                                ptr(partials), stream(A))
    _check(rc, 'pk_vocab_sample')


def vocab_reduce(partials, M, V, rows, mask, ids, pred, scores, need_lse):
    rc = load().pk_vocab_reduce(ptr(partials), M, V, ptr(rows), ptr(mask), ptr(ids), ptr(pred), ptr(scores),
                                1 if need_lse else 0, stream(partials))
    _check(rc, 'pk_vocab_reduce')


def vocab_ce(dtype, partials, M, V, A, W, bias, D, targets, rows, loss, lse=None):
    """loss[m] = lse(logits[m]) - logits[m][targets[rows[m]]] from the partials of vocab_sample(..., need_lse=True); lse (M,) f32 optional"""
    rc = load().pk_vocab_ce(dtype, ptr(partials), M, V, ptr(A), A.stride(-2), ptr(W), W.stride(0), f32p(bias, 'to_logits.bias'), D,
                            ptr(targets), ptr(rows), ptr(loss), ptr(lse), stream(A))
    _check(rc, 'pk_vocab_ce')


def ce_grad_slab(logits, lse, targets, rows, M, Vs, v0, scale, g, gT, db=None, scale_dev=None):
    """g / gT <- (softmax - onehot) * scale [* scale_dev[0]] of one slab of vocabulary columns (see the header); logits (M, >= Vs) f32"""
    rc = load().pk_ce_grad_slab(1 if g.dtype == torch.bfloat16 else 0, ptr(logits), logits.stride(0), ptr(lse), ptr(targets), ptr(rows), M, Vs, v0,
                                float(scale), ptr(scale_dev), ptr(g), g.stride(0), ptr(gT), gT.stride(0), ptr(db), stream(logits))
    _check(rc, 'pk_ce_grad_slab')


def topk_mask(scores, B, n, k, mask_id, mask, ids, rows_out=None, scores_next=None):
    rc = load().pk_topk_mask(ptr(scores), B, n, k, mask_id, ptr(mask), ptr(ids), ptr(rows_out), ptr(scores_next), stream(scores))
    _check(rc, 'pk_topk_mask')


def critic_head(x, w, b, D, nb, n_tot, n_prime, has_null, scale, u, noise_mult, out, seed=0, seed_dev=None):
    rc = load().pk_critic_head(ptr(x), x.stride(-2), f32p(w, 'critic head weight'), f32p(b, 'critic head bias'), D, nb, n_tot, n_prime, 1 if has_null else 0, scale,
                               ptr(u), noise_mult, seed & 0xFFFFFFFFFFFFFFFF, ptr(seed_dev), ptr(out), stream(x))
    _check(rc, 'pk_critic_head')


# ----------------------------------------------------------------------------- training-step kernels (csrc/train.hip, attn_train.hip)

def kind_of(dtype):
    """pk_pack's `kind` of the W-side operand image of compute dtype `dtype`"""
    return {F32: 0, BF16: 1, BF16X3: 2}[dtype]


def pack(src, R, K, transpose, out, Kp, kind, rows=None):
    """out[r][k] = src[k][r] (transpose) / src[r][k] for r < R, k < K; zero for K <= k < Kp.  rows: int32 gather of SOURCE rows."""
    rc = load().pk_pack(ptr(src), src.stride(-2), ptr(rows), R, K, 1 if transpose else 0, ptr(out), out.stride(-2), Kp, kind, stream(src))
    _check(rc, 'pk_pack')
    return out


class PackJob(ctypes.Structure):
    """include/phenaki_hip.h PkPackJob"""
    _fields_ = [('src', ctypes.c_void_p), ('out', ctypes.c_void_p), ('lds', ctypes.c_longlong), ('ldo', ctypes.c_longlong), ('R', ctypes.c_int),
                ('K', ctypes.c_int), ('Kp', ctypes.c_int), ('flags', ctypes.c_int), ('tile0', ctypes.c_int), ('tiles_x', ctypes.c_int)]


class SumJob(ctypes.Structure):
    """include/phenaki_hip.h PkSumJob"""
    _fields_ = [('src', ctypes.c_void_p), ('out', ctypes.c_void_p), ('stride', ctypes.c_longlong), ('E4', ctypes.c_longlong), ('S', ctypes.c_int),
                ('blk0', ctypes.c_int)]


def pack_job(src, R, K, transpose, out, Kp, kind, *, src_ld=None, out_ptr=None, out_ld=None):
    """one job of pack_multi / PackTable: out (R rows, Kp columns) = src or src^T (the OUTPUT has R rows and K data columns).  out_ptr / out_ld:
    a block inside a larger image (byte address of its first element, row pitch in elements)."""
    return PackJob(src.data_ptr(), out.data_ptr() if out_ptr is None else out_ptr, src.stride(-2) if src_ld is None else src_ld,
                   out.stride(-2) if out_ld is None else out_ld, R, K, Kp, (1 if transpose else 0) | (kind << 1), 0, 0)


def pack_multi(jobs, like):
    """up to 8 pack jobs per launch (more: several launches)"""
    for i in range(0, len(jobs), 8):
        chunk = jobs[i:i + 8]
        arr = (PackJob * len(chunk))(*chunk)
        _check(load().pk_pack_multi(ctypes.addressof(arr), len(chunk), stream(like)), 'pk_pack_multi')


class PackTable:
    """a job list kept in device memory and replayed with ONE launch (`run`): the operand images of a module's weights, re-packed every step"""

    def __init__(self, jobs, device):
        arr = (PackJob * len(jobs))(*jobs)
        tiles = load().pk_pack_table_prepare(ctypes.addressof(arr), len(jobs))
        if tiles < 0:
            _check(int(tiles), 'pk_pack_table_prepare')
        self.count, self.tiles = len(jobs), int(tiles)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)

    def run(self):
        _check(load().pk_pack_table(ptr(self.table), self.count, self.tiles, stream(self.table)), 'pk_pack_table')


class ColsumJob(ctypes.Structure):
    """include/phenaki_hip.h PkColsumJob"""
    _fields_ = [('src', ctypes.c_void_p), ('out', ctypes.c_void_p), ('ld', ctypes.c_longlong), ('M', ctypes.c_int), ('N', ctypes.c_int), ('blk0', ctypes.c_int),
                ('scale', ctypes.c_float)]


def colsum_multi(jobs):
    """jobs: [(src (M, N) f32 rows, M, N, out (N,))]: out[c] = sum_r src[r][c] for each, up to 8 per launch (M <= 8192, N % 4 == 0)"""
    for i in range(0, len(jobs), 8):
        chunk = jobs[i:i + 8]
        arr = (ColsumJob * len(chunk))(*[ColsumJob(s_.data_ptr(), o.data_ptr(), s_.stride(-2), M, N, 0, 1.0) for s_, M, N, o in chunk])
        _check(load().pk_colsum_multi(ctypes.addressof(arr), len(chunk), stream(chunk[0][0])), 'pk_colsum_multi')


def reduce_multi(sum_jobs, col_jobs):
    """the K-slice sums (sum_batch_multi's job tuples) and the short column sums (colsum_multi's) of a backward block in ONE launch"""
    if len(sum_jobs) > 8 or len(col_jobs) > 8 or not sum_jobs or not col_jobs:
        if sum_jobs:
            sum_batch_multi(sum_jobs)
        if col_jobs:
            colsum_multi(col_jobs)
        return
    assert all(E % 4 == 0 for _, _, _, E in sum_jobs)
    sa = (SumJob * len(sum_jobs))(*[SumJob(p.data_ptr(), o.data_ptr(), p.stride(0), E // 4, S, 0) for p, S, o, E in sum_jobs])
    ca = (ColsumJob * len(col_jobs))(*[ColsumJob(s_.data_ptr(), o.data_ptr(), s_.stride(-2), M, N, 0, 1.0) for s_, M, N, o in col_jobs])
    _check(load().pk_reduce_multi(ctypes.addressof(sa), len(sum_jobs), ctypes.addressof(ca), len(col_jobs), stream(sum_jobs[0][0])), 'pk_reduce_multi')


def colsum_deferred(src, M, N, out, defer):
    """pk_colsum now, or -- when the caller collects the column sums of a whole backward block (defer: a list) and the shape fits the one-launch
    form -- queued for ONE colsum_multi launch"""
    if defer is not None and M <= 8192 and N % 4 == 0 and src.stride(-2) % 4 == 0:
        defer.append((src, M, N, out))
        return out
    return colsum(src, M, N, out)


def sum_batch_multi(jobs):
    """jobs: [(part (S, >= E) f32, S, out, E)]: out[e] = sum_s part[s][e], up to 8 per launch"""
    for i in range(0, len(jobs), 8):
        chunk = jobs[i:i + 8]
        arr = (SumJob * len(chunk))(*[SumJob(p.data_ptr(), o.data_ptr(), p.stride(0), E // 4, S, 0) for p, S, o, E in chunk])
        assert all(E % 4 == 0 for _, _, _, E in chunk)
        _check(load().pk_sum_batch_multi(ctypes.addressof(arr), len(chunk), stream(chunk[0][0])), 'pk_sum_batch_multi')


def scatter_rows(src, rows, dst, M, D):
    rc = load().pk_scatter_rows(ptr(src), src.stride(-2), ptr(rows), ptr(dst), dst.stride(-2), M, D, stream(src))
    _check(rc, 'pk_scatter_rows')


def colsum(src, M, N, out, *, scale=1.0, accumulate=False, ld=None):
    """out[c] (+)= scale * sum_r src[r][c] (deterministic two-stage reduction)"""
    P = load().pk_colsum_parts(M)
    work = torch.empty((P * N,), device=src.device, dtype=torch.float32)
    rc = load().pk_colsum(ptr(src), src.stride(-2) if ld is None else ld, M, N, float(scale), ptr(out), 1 if accumulate else 0, ptr(work), stream(src))
    _check(rc, 'pk_colsum')
    return out


def layernorm_bwd(x, gamma, dy, dx, M, D, *, add=None, want_beta=False, eps=1e-5, defer=None):
    """dx = [add +] LN backward; returns (dgamma, dbeta | None).  defer (list): the column sum of the partials is queued there (colsum_multi)"""
    P = load().pk_ln_bwd_parts(M)
    # dgamma | dbeta partials as the two halves of one (P, 2 D) buffer (the kernel recognises pb == pg + D): one column sum finishes both
    pgb = torch.empty((P, 2 * D if want_beta else D), device=x.device, dtype=torch.float32)
    rc = load().pk_layernorm_bwd(ptr(x), x.stride(-2), f32p(gamma, 'LayerNorm gamma'), ptr(dy), dy.stride(-2), ptr(add), add.stride(-2) if add is not None else 0,
                                 ptr(dx), dx.stride(-2), ptr(pgb), pgb.data_ptr() + 4 * D if want_beta else None, eps, M, D, stream(x))
    _check(rc, 'pk_layernorm_bwd')
    out = colsum_deferred(pgb, P, pgb.shape[1], torch.empty((pgb.shape[1],), device=x.device, dtype=torch.float32), defer)
    return (out[:D], out[D:]) if want_beta else (out, None)


def geglu(h, goff, out, M, F):
    rc = load().pk_geglu(ptr(h), h.stride(-2), goff, ptr(out), out.stride(-2), M, F, stream(h))
    _check(rc, 'pk_geglu')


def geglu_bwd(h, goff, dout, dh, M, F):
    rc = load().pk_geglu_bwd(ptr(h), h.stride(-2), goff, ptr(dout), dout.stride(-2), ptr(dh), dh.stride(-2), M, F, stream(h))
    _check(rc, 'pk_geglu_bwd')


def scaled_diff(a, b, scale, out, scale_dev=None):
    """out = (a - b) * scale (* scale_dev[0]) over contiguous f32 tensors of the same size"""
    rc = load().pk_scaled_diff(f32p(a, 'a'), f32p(b, 'b'), float(scale), ptr(scale_dev), ptr(out), a.numel(), stream(a))
    _check(rc, 'pk_scaled_diff')
    return out


def mul(a, b, out):
    rc = load().pk_mul(f32p(a, 'a'), f32p(b, 'b'), ptr(out), a.numel(), stream(a))
    _check(rc, 'pk_mul')
    return out


def sign(z, out, value=1.0):
    rc = load().pk_sign(f32p(z, 'z'), float(value), ptr(out), z.numel(), stream(z))
    _check(rc, 'pk_sign')
    return out


def leaky_bwd(y, dy, dz, M, N, slope=0.1):
    rc = load().pk_leaky_bwd(ptr(y), y.stride(-2), ptr(dy), dy.stride(-2), ptr(dz), dz.stride(-2), M, N, slope, stream(y))
    _check(rc, 'pk_leaky_bwd')


def peg_bwd(dy, x, wt, dx, B, T, H, W, D, causal, want_wgrad=True, defer=None):
    """dx = dy + transposed stencil; returns the (27, D) tap gradient (or None)"""
    rows = B * T * H * W
    part = None
    if want_wgrad:
        P = load().pk_peg_wgrad_parts(rows)
        part = torch.empty((P, 27 * D), device=dy.device, dtype=torch.float32)
    rc = load().pk_peg_bwd(ptr(dy), ptr(x), f32p(wt, 'PEG taps'), ptr(dx), ptr(part), B, T, H, W, D, 1 if causal else 0, stream(dy))
    _check(rc, 'pk_peg_bwd')
    if part is None:
        return None
    return colsum_deferred(part, part.shape[0], 27 * D, torch.empty((27 * D,), device=dy.device, dtype=torch.float32), defer).view(27, D)


def embed_bwd(dy, ids, alpha, dtok, dpos, S, n, D):
    rc = load().pk_embed_bwd(ptr(dy), ptr(ids), float(alpha), ptr(dtok), ptr(dpos), S, n, D, stream(dy))
    _check(rc, 'pk_embed_bwd')


def bias_gather(tab, code, off, out, heads, n):
    rc = load().pk_bias_gather(ptr(tab), tab.stride(0), ptr(code), off, ptr(out), heads, n, stream(tab))
    _check(rc, 'pk_bias_gather')


def bias_scatter(dbias, code, off, dtab, heads, n):
    rc = load().pk_bias_scatter(ptr(dbias), ptr(code), off, ptr(dtab), dtab.stride(0), heads, n, stream(dbias))
    _check(rc, 'pk_bias_scatter')


def sum_batch(src, S, out, E):
    rc = load().pk_sum_batch(ptr(src), src.stride(0), S, ptr(out), E, stream(src))
    _check(rc, 'pk_sum_batch')


def bce_head(e, w, b, labels, M, D, *, scale=0.0, logits=None, loss_rows=None, de=None, scale_dev=None):
    """returns (dw (D,), db (1,)) when de is given, else None"""
    pw = pb = None
    P = load().pk_ln_bwd_parts(M)
    if de is not None:
        pw = torch.empty((P, D), device=e.device, dtype=torch.float32)
        pb = torch.empty((P, 1), device=e.device, dtype=torch.float32)
    rc = load().pk_bce_head(ptr(e), e.stride(-2), f32p(w, 'critic head weight'), f32p(b, 'critic head bias'), ptr(labels), float(scale), ptr(scale_dev), ptr(logits), ptr(loss_rows),
                            ptr(de), de.stride(-2) if de is not None else 0, ptr(pw), ptr(pb), M, D, stream(e))
    _check(rc, 'pk_bce_head')
    if de is None:
        return None
    dw = colsum(pw, P, D, torch.empty((D,), device=e.device, dtype=torch.float32))
    db = colsum(pb, P, 1, torch.empty((1,), device=e.device, dtype=torch.float32))
    return dw, db


ATTN_PREP_BWD_PARTS = 1024


def attn_train_prep(q, kv, null_kv, q_scale, k_scale, scale, Qh, Kh, Vh, S, h, n, n_kv, nnull):
    rc = load().pk_attn_train_prep(ptr(q), q.stride(-2), ptr(kv), kv.stride(-2), f32p(null_kv, 'null_kv') if nnull else None, f32p(q_scale, 'q_scale'),
                                   f32p(k_scale, 'k_scale'), scale, ptr(Qh), ptr(Kh), ptr(Vh), S, h, n, n_kv, nnull, stream(q))
    _check(rc, 'pk_attn_train_prep')


def attn_train_prep_bwd(q, kv, null_kv, q_scale, k_scale, scale, dQh, dKh, dVh, dq, dkv, S, h, n, n_kv, nnull, defer=None):
    """-> (dq_scale (64,), dk_scale (64,), dnull_kv | None); dq / dkv are overwritten"""
    dev = q.device
    pqk = torch.empty((ATTN_PREP_BWD_PARTS, 128), device=dev, dtype=torch.float32)     # dq_scale | dk_scale partials (the kernel recognises pk == pq + 64)
    dnull = torch.empty((h, 2 * nnull, 64), device=dev, dtype=torch.float32) if nnull else None
    rc = load().pk_attn_train_prep_bwd(ptr(q), q.stride(-2), ptr(kv), kv.stride(-2), f32p(null_kv, 'null_kv') if nnull else None, f32p(q_scale, 'q_scale'),
                                       f32p(k_scale, 'k_scale'), scale, ptr(dQh), ptr(dKh), ptr(dVh), ptr(dq), dq.stride(-2), ptr(dkv), dkv.stride(-2),
                                       ptr(pqk), pqk.data_ptr() + 256, ptr(dnull), S, h, n, n_kv, nnull, stream(q))
    _check(rc, 'pk_attn_train_prep_bwd')
    out = colsum_deferred(pqk, ATTN_PREP_BWD_PARTS, 128, torch.empty((128,), device=dev, dtype=torch.float32), defer)
    return out[:64], out[64:], dnull


def attn_bwd(Qh, Kh, Vh, O, dO, dQh, dKh, dVh, S, h, n, n_kv, nnull, *, bias=None, kmask=None, dS=None, slopes=None, causal=False, split_bf16=False, lse=None,
             bf16_products=False):
    """lse ((S h n,) f32 from attn_fwd(lse=...)): the backward skips its own log-sum-exp pass.  bf16_products: single bf16 MFMA products (the bf16 mode)"""
    dev = Qh.device
    flags = (1 if split_bf16 else 0) | (2 if lse is not None else 0) | (4 if bf16_products else 0)
    if lse is None:
        lse = torch.empty((S * h * n,), device=dev, dtype=torch.float32)
    drow = torch.empty((S * h * n,), device=dev, dtype=torch.float32)
    nwork = load().pk_attn_bwd_work(S, h, n, n_kv, nnull)                  # few key tiles: partial dK / dV slabs of the query-tile groups
    work = torch.empty((nwork,), device=dev, dtype=torch.float32) if nwork > 0 else None
    rc = load().pk_attn_bwd_ws(ptr(Qh), ptr(Kh), ptr(Vh), ptr(O), O.stride(-2), 1 if O.dtype == torch.bfloat16 else 0, ptr(dO), dO.stride(-2), ptr(bias), ptr(kmask),
                               f32p(slopes, 'ALiBi slopes') if causal else None, 1 if causal else 0, ptr(dQh), ptr(dKh), ptr(dVh), ptr(dS), ptr(lse), ptr(drow), S, h, n, n_kv, nnull, flags,
                               ptr(work), nwork if nwork > 0 else 0, stream(Qh))
    _check(rc, 'pk_attn_bwd')


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step):
    rc = load().pk_adamw(ptr(p), ptr(g), ptr(m), ptr(v), float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step), p.numel(), stream(p))
    _check(rc, 'pk_adamw')


def adamw_multi(entries, lr, beta1, beta2, eps, wd, step, device):
    """entries: list of (p, g, m, v) contiguous f32 device tensors updated with one hyper-parameter set and step number"""
    import numpy as np
    table = np.empty((len(entries), 5), dtype=np.int64)
    for i, (p, g, m, v) in enumerate(entries):
        table[i] = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
    rc = load().pk_adamw_multi(table.ctypes.data, len(entries), float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step),
                               torch.cuda.current_stream(device).cuda_stream)
    _check(rc, 'pk_adamw_multi')


def gemm_splitk(dtype, A, W, M, N, K, splits, C, bias=None, tile=0):
    """C (splits, M, N) f32 <- the K-slices of A @ W^T (see the header; bias on slice 0; tile 1: 128 x 128 tiles); reduce with sum_batch"""
    rc = load().pk_gemm_splitk(dtype, ptr(A), A.stride(-2), ptr(W), W.stride(0), M, N, K, splits, ptr(C), N, f32p(bias, 'bias'), int(tile), stream(C))
    _check(rc, 'pk_gemm_splitk')


# ----------------------------------------------------------------------------- discriminator layout kernels (csrc/conv.hip)

def conv_out_size(n, k, stride, pad):
    return (n + 2 * pad - k) // stride + 1


def im2col(x, B, H, W, C, kh, kw, stride, pad, cols):
    """x (B H W, C) channels-last pixel rows -> cols (B Ho Wo, kh kw C), column (ky, kx, c)"""
    _check(load().pk_im2col(ptr(x), B, H, W, C, kh, kw, stride, pad, ptr(cols), cols.stride(0), stream(x)), 'pk_im2col')
    return cols


def col2im(cols, B, H, W, C, kh, kw, stride, pad, dx):
    """the adjoint of im2col: dx (B H W, C) <- sum of the patch-matrix entries that read each pixel"""
    _check(load().pk_col2im(ptr(cols), cols.stride(0), B, H, W, C, kh, kw, stride, pad, ptr(dx), stream(cols)), 'pk_col2im')
    return dx


def nchw_to_rows(img, Cp, rows):
    B, C, H, W = img.shape
    _check(load().pk_nchw_to_rows(ptr(img), B, C, H, W, Cp, ptr(rows), stream(img)), 'pk_nchw_to_rows')
    return rows


def rows_to_nchw(rows, Cp, img):
    B, C, H, W = img.shape
    _check(load().pk_rows_to_nchw(ptr(rows), B, C, H, W, Cp, ptr(img), stream(rows)), 'pk_rows_to_nchw')
    return img


def pick_frames(video, frame, img, place=False):
    """img (B, C, H, W) <- video[b, :, frame[b]] (cvivit.py:217-224), or the adjoint into a zeroed video when place"""
    B, C, F, H, W = video.shape
    # `frame` reaches the kernel as a raw `const int*`: an int64 tensor (what topk returns), a host tensor or a short one must fail here
    if frame.dtype != torch.int32 or frame.device != video.device or not frame.is_contiguous() or frame.numel() != B:
        raise RuntimeError(f'pick_frames: frame must be a contiguous int32 tensor of {B} indices on {video.device}, got {frame.dtype} '
                           f'{tuple(frame.shape)} on {frame.device}')
    _check(load().pk_pick_frames(ptr(video), ptr(frame), B, C, F, H, W, ptr(img), 1 if place else 0, stream(video)), 'pk_pick_frames')
    return video if place else img


def bmm(A, B, C, tA, tB, batch, M, N, K, *, lda, ldb, ldc, sA=0, sB=0, sC=0, accumulate=False):
    """C[z] = op(A[z]) op(B[z]) in exact f32 (any shape / stride); op = transpose when tA / tB"""
    _check(load().pk_bmm(ptr(A), lda, sA, 1 if tA else 0, ptr(B), ldb, sB, 1 if tB else 0, ptr(C), ldc, sC, batch, M, N, K,
                         1 if accumulate else 0, stream(A)), 'pk_bmm')
    return C


def row_softmax(a, b, c, out, mode):
    """rows of the last dimension (contiguous f32): mode 0 softmax(a); 1 a (b - <a, b>); 2 c (b - <a, b>) - b <c, a>  (see the header)"""
    n = a.shape[-1]
    _check(load().pk_row_softmax(ptr(a), ptr(b), ptr(c), ptr(out), a.numel() // n, n, mode, stream(a)), 'pk_row_softmax')
    return out


def row_l2scale(x, sc, dz, gx, gsc, o0, o1, o2, mode):
    d = x.shape[-1]
    _check(load().pk_row_l2scale(ptr(x), ptr(sc), ptr(dz), ptr(gx), ptr(gsc), ptr(o0), ptr(o1), ptr(o2), x.numel() // d, d, mode, stream(x)), 'pk_row_l2scale')


def row_ln_bwd2(x, gamma, dy, u, w, eps, grad_x, grad_gamma_rows, grad_dy):
    D = x.shape[-1]
    _check(load().pk_row_ln_bwd2(ptr(x), ptr(gamma), ptr(dy), ptr(u), ptr(w), float(eps), ptr(grad_x), ptr(grad_gamma_rows), ptr(grad_dy),
                                 x.numel() // D, D, stream(x)), 'pk_row_ln_bwd2')


class TorchPhilox:
    """where torch's device generator stands for ONE `uniform_` fill of `numel` float32 elements (ATen's launch geometry on this device):
    seed, Philox offset, thread stride; `.advance()` moves the generator past the fill exactly as the real op would"""

    def __init__(self, device, numel):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self.gen = torch.cuda.default_generators[idx]
        props = torch.cuda.get_device_properties(idx)
        blocks = min(props.multi_processor_count * (props.max_threads_per_multi_processor // 256), (numel + 255) // 256)
        self.stride = 256 * blocks
        self.seed = self.gen.initial_seed() & 0xFFFFFFFFFFFFFFFF
        self.offset = self.gen.get_offset()
        self.increment = ((numel - 1) // (self.stride * 4) + 1) * 4

    def advance(self):
        self.gen.set_offset(self.offset + self.increment)


def vocab_sample_philox(dtype, A, W, bias, M, V, D, temperature, rows, spec, need_lse, partials):
    rc = load().pk_vocab_sample_philox(dtype, ptr(A), A.stride(-2), ptr(W), W.stride(0), f32p(bias, 'to_logits.bias'), M, V, D, temperature, ptr(rows),
                                       spec.seed, spec.offset, spec.stride, 1 if need_lse else 0, ptr(partials), stream(A))
    _check(rc, 'pk_vocab_sample_philox')
