"""`torch.library` registration of the core kernels ("Python wrappers register them as torch.library custom ops
(phenaki_mi355x::patch_embed, ...) so they compose with torch.no_grad / streams").

    import phenaki_pytorch_amd.ops                    # registers the namespace once
    torch.ops.phenaki_mi355x.gemm(x, w_packed, bias, res, out, dtype, act)

The modules of this package call the ctypes wrappers of `_lib.py` directly (one Python frame less per launch on a launch-bound path);
these ops are the SAME C-ABI entry points behind dispatcher schemas, for callers that want them as graph-visible operators: every op
writes into caller-allocated outputs (`Tensor(a!)` arguments, returned for chaining), runs on the current HIP stream, allocates nothing and
has no CPU implementation (a CPU tensor raises: there is no fallback).  Schemas are registered at import without touching the GPU.
"""
import torch

from . import _lib as L

NAMESPACE = 'phenaki_mi355x'
_lib = torch.library.Library(NAMESPACE, 'DEF')

_lib.define('gemm(Tensor x, Tensor w_packed, Tensor? bias, Tensor? res, Tensor(a!) out, int dtype, int act) -> Tensor(a!)')
_lib.define('layernorm(Tensor x, Tensor gamma, Tensor? beta, Tensor(a!) out, float eps) -> Tensor(a!)')
_lib.define('peg(Tensor x, Tensor taps, Tensor bias, Tensor(a!) out, int[] shape, bool causal) -> Tensor(a!)')
_lib.define('attn_prep(Tensor q, Tensor kv, Tensor? null_kv, Tensor q_scale, Tensor k_scale, float scale, Tensor(a!) qp, Tensor(b!) kp, Tensor(c!) vt, '
            'int dtype, int S, int heads, int n, int n_kv) -> ()')
_lib.define('attn_fwd(Tensor qp, Tensor kp, Tensor vt, Tensor? bias, Tensor? kmask, Tensor(a!) out, int dtype, int S, int heads, int n, int n_kv, '
            'int nnull) -> Tensor(a!)')
_lib.define('patch_embed(Tensor video, Tensor w_folded, Tensor s, Tensor t, Tensor(a!) out, int patch_h, int patch_w, int first_frame, '
            'int temporal_patches, int frames_per_patch, float eps) -> Tensor(a!)')
_lib.define('vocab_sample(Tensor x, Tensor w_packed, Tensor bias, Tensor? noise, Tensor? rows, Tensor(a!) partials, int dtype, float temperature, '
            'int seed, bool need_lse) -> Tensor(a!)')
_lib.define('vocab_reduce(Tensor partials, Tensor? rows, Tensor? mask, Tensor? ids, Tensor(a!) pred, Tensor? scores, int vocab, bool need_lse) -> Tensor(a!)')
_lib.define('lfq_decode(Tensor ids, Tensor project_out_weight, Tensor project_out_bias, Tensor(a!) out) -> Tensor(a!)')


def _gemm(x, w_packed, bias, res, out, dtype, act):
    M, K = x.shape
    L.gemm(dtype, x, w_packed, M, out.shape[1] * (2 if act == L.ACT_GEGLU else 1), K, C=out, bias=bias, res=res, act=act)
    return out


def _layernorm(x, gamma, beta, out, eps):
    M, D = x.shape
    if out.dtype == torch.float32:
        L.layernorm(x, gamma, beta, M, D, out2=out, eps=eps)
    else:
        L.layernorm(x, gamma, beta, M, D, out=out, eps=eps)
    return out


def _peg(x, taps, bias, out, shape, causal):
    b, t, h, w = shape
    L.peg(x, taps, bias, out, b, t, h, w, x.shape[1], causal)
    return out


def _attn_prep(q, kv, null_kv, q_scale, k_scale, scale, qp, kp, vt, dtype, S, heads, n, n_kv):
    nnull = null_kv.shape[1] // 2 if null_kv is not None else 0
    L.attn_prep(dtype, q, kv, null_kv, q_scale, k_scale, scale, qp, kp, vt, S, heads, n, n_kv, nnull)


def _attn_fwd(qp, kp, vt, bias, kmask, out, dtype, S, heads, n, n_kv, nnull):
    L.attn_fwd(dtype, qp, kp, vt, out, S, heads, n, n_kv, nnull, bias=bias, kmask=kmask)
    return out


def _patch_embed(video, w_folded, s, t, out, patch_h, patch_w, first_frame, temporal_patches, frames_per_patch, eps):
    L.patch_embed(video, patch_h, patch_w, out.shape[1], [(w_folded, s, t, out, first_frame, temporal_patches, frames_per_patch)], eps=eps)
    return out


def _vocab_sample(x, w_packed, bias, noise, rows, partials, dtype, temperature, seed, need_lse):
    M = rows.numel() if rows is not None else x.shape[0]
    L.vocab_sample(dtype, x, w_packed, bias, M, w_packed.shape[0], x.shape[1], temperature, noise, rows, seed, need_lse, partials)
    return partials


def _vocab_reduce(partials, rows, mask, ids, pred, scores, vocab, need_lse):
    M = rows.numel() if rows is not None else pred.numel()
    L.vocab_reduce(partials, M, vocab, rows, mask, ids, pred, scores, need_lse)
    return pred


def _lfq_decode(ids, w, b, out):
    L.lfq_decode(ids.reshape(-1), w, b, out, out.shape[0], out.shape[1], w.shape[1])
    return out


for _name, _fn in (('gemm', _gemm), ('layernorm', _layernorm), ('peg', _peg), ('attn_prep', _attn_prep), ('attn_fwd', _attn_fwd),
                   ('patch_embed', _patch_embed), ('vocab_sample', _vocab_sample), ('vocab_reduce', _vocab_reduce), ('lfq_decode', _lfq_decode)):
    _lib.impl(_name, _fn, 'CUDA')                      # HIP devices dispatch under the CUDA key; no CPU kernel is registered

OPS = ('gemm', 'layernorm', 'peg', 'attn_prep', 'attn_fwd', 'patch_embed', 'vocab_sample', 'vocab_reduce', 'lfq_decode')
