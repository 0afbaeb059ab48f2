"""The training step on the MI355X kernels: `Phenaki.forward` with gradients -- reference
/root/reference/phenaki_pytorch/phenaki_pytorch.py:562-687 under autograd (caller phenaki_trainer.py:351-388).

    loss = phenaki(videos, texts=...)          # grad mode on, trainable parameters -> phenaki_loss() below
    loss.backward()                            # MaskGit (CE of the masked tokens) and critic (BCE) parameters get .grad

Every block of the trunk is ONE torch.autograd.Function whose forward and backward are calls into the C ABI (no ATen arithmetic):

    _Embed          token + position embedding (+ gradient_shrink_alpha)       pk_embed            / pk_embed_bwd
    _PEGBlock       x + dsconv(x)                                              pk_peg              / pk_peg_bwd (+ pk_colsum)
    _AttnBlock      x + to_out(attention(norm(x) [, context]))                 pk_layernorm, pk_gemm, pk_attn_prep, pk_attn_fwd
                                                                               / pk_attn_train_prep, pk_attn_bwd, pk_attn_train_prep_bwd,
                                                                                 pk_pack + pk_gemm (dX, dW), pk_layernorm_bwd
    _FFBlock        x + W2 geglu(W1 LayerNorm(x))                              pk_layernorm, pk_gemm, pk_geglu / pk_geglu_bwd, ...
    _LayerNormFn    norm_out                                                   pk_layernorm        / pk_layernorm_bwd
    _PositionBias   ContinuousPositionBias as (heads, n, n)                    relative-position-table MLP + pk_bias_gather / pk_bias_scatter
    _VocabCrossEntropy   to_logits + cross entropy on the masked rows          pk_vocab_sample, pk_vocab_ce / pk_ce_grad_slab + pk_gemm
    _BCEHead        critic to_logits / to_pred + BCE-with-logits               pk_bce_head

Activations and gradients are f32 in HBM (288 GB: every block keeps what its backward needs instead of recomputing it); the GEMMs run in the
module's compute dtype ('fp32' | 'bf16x3' | 'bf16').  Matrix products of the backward pass: dX = dY W through pk_gemm on pk_pack(W^T),
dW = dY^T X through pk_gemm on pk_pack(dY^T) and pk_pack(X^T) (contraction over the rows, zero-padded to the k-tile).
Round 6 (launch count): the W / W^T images of a whole Transformer are persistent and refreshed by ONE launch per pass (`WeightImages`,
pk_pack_table); the activation transposes of a backward block leave in one launch (`transposes`, pk_pack_multi), its K-slice sums and column sums
in another (L.reduce_multi); the forward attention hands its log-sum-exp to the backward kernels (pk_attn_fwd_lse); the critic's gumbel sample and
the cross entropy share one pass over the vocabulary (`_VocabCrossEntropy(shared=...)`).
Limits (asserted): dropout 0 (the reference default).  The tokenizer's own reconstruction step is train_cvivit.py.
"""
import math
import os
import weakref

import torch

from . import _lib as L
from .attention import (Attention, FeedForwardSeq, LayerNorm, PEG, compute_dtype_of, exists, pack_linear_weight, resolve_dtype, round_up)


# vocabulary columns per step of the cross-entropy backward (a multiple of 64; tuning knob PK_CE_SLAB)
CE_SLAB = int(os.environ.get('PK_CE_SLAB', '8192'))


# bf16 mode: the attention backward's tile products as single bf16 MFMAs on the operands the bf16 forward kernels multiply (0: split-bf16 products of the f32
# operands, and a log-sum-exp pass of their own); knob PK_ATTN_BWD_BF16, DESIGN 5.1
ATTN_BWD_BF16 = os.environ.get('PK_ATTN_BWD_BF16', '1') != '0'


def _q(dtype):
    return 64 if dtype == L.BF16 else 32


def _f32(shape, dev):
    return torch.empty(shape, device=dev, dtype=torch.float32)


def _zeros(shape, dev):
    return torch.zeros(shape, device=dev, dtype=torch.float32)


def _operand(x, dtype):
    """(rows, K) f32 -> the `W`-side GEMM operand image of compute dtype `dtype` (K zero-padded to the k-tile)"""
    return pack_linear_weight(x, dtype)


def pack_operand(src, dtype, *, transpose=False, side='w', rows=None, nrows=None):
    """f32 matrix -> operand image of a pk_gemm call in compute dtype `dtype`, contraction index zero-padded to the k-tile.
    side 'w': the weight-side image (f32 | bf16 | split-bf16 planes); side 'a': the activation side (bf16 in bf16 mode, else f32).
    transpose: the image holds src^T.  rows (int32): gather of source rows (nrows = how many)."""
    R0 = src.shape[0] if rows is None else nrows
    C0 = src.shape[1]
    R, K = (C0, R0) if transpose else (R0, C0)
    Kp = round_up(K, _q(dtype))
    kind = L.kind_of(dtype) if side == 'w' else (1 if dtype == L.BF16 else 0)
    out = torch.empty((R, Kp), device=src.device, dtype=torch.bfloat16 if kind == 1 else torch.float32)
    return L.pack(src, R, K, transpose, out, Kp, kind, rows=rows)


def linear_fwd(dtype, x, W, *, bias=None, res=None, act=L.ACT_NONE, Wimg=None):
    """x (M, K) f32 @ W (N, K)^T [+ bias] [+ res] -> (M, N) f32.  Wimg: the operand image of W when the caller holds one (WeightImages)"""
    M, K = x.shape
    N = W.shape[0]
    y = _f32((M, N), x.device)
    L.gemm(dtype, x, Wimg if Wimg is not None else pack_operand(W, dtype), M, N, K, C=y, bias=bias, res=res, act=act)
    return y


def _weight_grad_gemm(dtype, dyT, xT, N, K, Mp, dW, defer=None):
    """dW (N, K) = dyT (N, Mp) @ xT (K, Mp)^T: the contraction runs over the rows of the batch (Mp = 4608 at B = 8) while the output has only
    ceil(N / 64) ceil(K / 64) tiles -- 64 for a 512 x 512 projection on 256 CUs -- so the product is cut into K-slices (pk_gemm_splitk) that are
    added in index order (pk_sum_batch: deterministic).  defer (list): the slice sum is queued there instead of launched -- the caller adds the
    slices (and the column sums) of a whole backward block in ONE launch, `L.reduce_multi`."""
    q = _q(dtype)
    tiles = ((N + 63) // 64) * ((K + 63) // 64)
    splits = 1
    if K % 4 == 0 and dW.is_contiguous():
        # up to 64 slices (the C ABI's limit) for the discriminator's convolution weights: a 64 x 576 gradient contracted over 524 288 pixel rows
        # is 9 tiles -- with 8 slices each workgroup walked 2 048 k-tiles (2.2 ms); the trunk's shapes (>= 64 tiles) still resolve to <= 8
        for cand in (64, 32, 16, 8, 4, 2):
            if tiles * cand <= 768 and Mp % (cand * q) == 0 and Mp // cand >= (16 if cand > 8 else 4) * q:
                splits = cand
                break
    if splits == 1:
        L.gemm(dtype, dyT, xT, N, K, Mp, C=dW)
        return
    part = torch.empty((splits, N * K), device=dW.device, dtype=torch.float32)
    L.gemm_splitk(dtype, dyT, xT, N, K, Mp, splits, part)
    if defer is not None and (N * K) % 4 == 0:
        defer.append((part, splits, dW, N * K))
    else:
        L.sum_batch(part, splits, dW, N * K)


def transposes(dtype, items, casts=()):
    """[(src (M, C) f32 rows, side 'a' | 'w')] -> [src^T as the (C, Mp) operand image of that side]: the activation transposes a backward block
    needs for its weight gradients, laid down by ONE pk_pack_multi launch (round 6: one pk_pack launch each before).
    casts: f32 row matrices that are also the A operand of a product of the block -- in the bf16 mode their bf16 copies are written by the same
    launch (the GEMM then takes the LDS-DMA main loop instead of the register-staged f32-A kernel); returned after the transposes, in order
    (the tensors themselves in the other modes, whose GEMMs read f32 A operands)."""
    q = _q(dtype)
    jobs, outs = [], []
    for src, side in items:
        M, C = src.shape
        Mp = round_up(M, q)
        kind = L.kind_of(dtype) if side == 'w' else (1 if dtype == L.BF16 else 0)
        out = torch.empty((C, Mp), device=src.device, dtype=torch.bfloat16 if kind == 1 else torch.float32)
        jobs.append(L.pack_job(src, C, M, True, out, Mp, kind))
        outs.append(out)
    for src in casts:
        if dtype == L.BF16 and src.dtype == torch.float32 and src.shape[1] % 8 == 0 and src.shape[0] >= 256:
            out = torch.empty(tuple(src.shape), device=src.device, dtype=torch.bfloat16)
            jobs.append(L.pack_job(src, src.shape[0], src.shape[1], False, out, src.shape[1], 1))
            outs.append(out)
        else:
            outs.append(src)
    if jobs:
        L.pack_multi(jobs, items[0][0] if items else casts[0])
    return outs


def a_operand(dtype, x):
    """the A-operand form of f32 rows x: a bf16 copy in the bf16 mode (one cast launch), x itself otherwise"""
    return transposes(dtype, [], [x])[0]


def linear_bwd(dtype, x, W, dy, *, need_dx=True, add=None, need_dw=True, dw_out=None, Wt=None, dyT=None, xT=None, defer=None, dyA=None):
    """gradients of y = x W^T: dx = dy W [+ add] (M, K), dW = dy^T x (N, K).  dw_out: preallocated (N, K) destination (may be a row slice).
    Wt: the (K, Kp(N)) operand image of W^T when the caller holds one (`WeightImages`); dyT / xT: the transposed operand images of `transposes`;
    defer: see _weight_grad_gemm.  dyA: dy as the A operand of dx = dy W (its bf16 copy in the bf16 mode, `transposes(..., casts=[dy])`)."""
    M, K = x.shape
    N = W.shape[0]
    dx = dW = None
    if need_dx:
        dx = _f32((M, K), x.device)
        L.gemm(dtype, dyA if dyA is not None else dy, Wt if Wt is not None else pack_operand(W, dtype, transpose=True), M, K, N, C=dx, res=add)
    if need_dw:
        Mp = round_up(M, _q(dtype))
        if dyT is None:
            dyT, xT = transposes(dtype, [(dy, 'a'), (x, 'w')])          # (N, Mp), (K, Mp)
        dW = dw_out if dw_out is not None else _f32((N, K), x.device)
        _weight_grad_gemm(dtype, dyT, xT, N, K, Mp, dW, defer)
    return dx, dW


class WeightImages:
    """The operand images of a module's projection weights in one compute dtype -- for every Linear both the forward image of W and the image of
    W^T its backward multiplies by -- held in persistent buffers and re-packed by ONE launch per pass (`refresh`, pk_pack_table).  Round 6: the
    training step issued ~220 single-matrix pk_pack launches of 5-9 us for these (every weight twice per step).  The pad rows / columns of an image
    are zeroed once at allocation and never written again.  Images are tied to the storage of the parameters they were built for (`key`)."""

    def __init__(self, dtype, device):
        self.dtype, self.device = dtype, device
        self.q, self.kind = _q(dtype), L.kind_of(dtype)
        self.td = torch.bfloat16 if self.kind == 1 else torch.float32
        self.jobs, self.table, self.params, self.live = [], None, [], []

    def _image(self, rows, cols):
        return torch.zeros((rows, round_up(cols, self.q)), device=self.device, dtype=self.td)

    def _job(self, src, R, K, transpose, out, Kp=None):
        """out (R rows): columns [0, K) from src / src^T, zero up to Kp (default: the whole padded row of a full image)"""
        self.jobs.append(L.pack_job(src, R, K, transpose, out, out.shape[-1] if Kp is None else Kp, self.kind))

    def both(self, w):
        """w (N, K) -> (image of W: (N, Kp(K)), image of W^T: (K, Kp(N)))"""
        self.live.append(w)
        w = w.detach()
        N, K = w.shape
        f, t = self._image(N, K), self._image(K, N)
        self._job(w, N, K, False, f)
        self._job(w, K, N, True, t)
        self.params.append(w)
        return f, t

    def feedforward(self, w1, w2):
        """the GEGLU feed-forward's layouts (see _FFBlock): w1p (2 Fp, Kp(D)) value rows [0, F) | gate rows [Fp, Fp + F); w1t (D, Kp(2 Fp)) its
        transpose; w2 (D, Kp(F)); w2t (Fp, Kp(D)) with zero pad rows"""
        self.live += [w1, w2]
        w1, w2 = w1.detach(), w2.detach()
        D, F = w2.shape
        Fp = round_up(F, 8)
        w1p, w1t = self._image(2 * Fp, D), self._image(D, 2 * Fp)
        self._job(w1[:F], F, D, False, w1p[:F])
        self._job(w1[F:], F, D, False, w1p[Fp:Fp + F])
        self._job(w1[:F], D, F, True, w1t[:, :Fp], Fp)
        self._job(w1[F:], D, F, True, w1t[:, Fp:2 * Fp], Fp)
        w2f, w2t = self._image(D, F), self._image(Fp, D)
        self._job(w2, D, F, False, w2f)
        self._job(w2, F, D, True, w2t[:F])
        self.params += [w1, w2]
        return dict(w1p=w1p, w1t=w1t, w2=w2f, w2t=w2t)

    def key(self):
        """storage identity of the LIVE parameters (a `.data = ...` / `.to()` moves them: the job table then points at dead storage)"""
        return tuple((p.data_ptr(), tuple(p.shape)) for p in self.live)

    def refresh(self):
        if len(self.jobs) <= 8:                                          # one block's weights (a direct caller): the jobs travel in the kernel arguments
            L.pack_multi(self.jobs, self.params[0])
            return
        if self.table is None:
            self.table = L.PackTable(self.jobs, self.device)
        self.table.run()


# module -> {compute dtype: WeightImages}; outside the module so that deepcopy (EMA, copy_for_eval) and state_dict never see the images
_IMAGES = weakref.WeakKeyDictionary()


def transformer_images(tr, dtype):
    """the WeightImages of a Transformer (attention.py:277-332), built on first use and rebuilt when a parameter's storage changed"""
    cache = _IMAGES.setdefault(tr, {})
    wi = cache.get(dtype)
    if wi is not None and wi._key == wi.key():
        return wi
    dev = next(tr.parameters()).device
    wi = WeightImages(dtype, dev)
    wi.layers = []
    for peg, self_attn, cross_attn, ff in tr.layers:
        rec = {}
        for name, at in (('sa', self_attn), ('ca', cross_attn)):
            rec[name] = None if at is None else dict(wq=wi.both(at.to_q.weight), wkv=wi.both(at.to_kv.weight), wo=wi.both(at.to_out.weight))
        rec['ff'] = wi.feedforward(ff[1].weight, ff[4].weight)
        wi.layers.append(rec)
    wi._key = wi.key()
    cache[dtype] = wi
    return wi


def linear_images(lin, dtype):
    """(image of W, image of W^T) of one Linear (the vocabulary head), same caching rule as transformer_images"""
    cache = _IMAGES.setdefault(lin, {})
    wi = cache.get(dtype)
    if wi is None or wi._key != wi.key():
        wi = WeightImages(dtype, lin.weight.device)
        wi.pair = wi.both(lin.weight)
        wi._key = wi.key()
        cache[dtype] = wi
    return wi


# ------------------------------------------------------------------------------------------------------------ blocks

class _LayerNormFn(torch.autograd.Function):
    """gamma-only LayerNorm (attention.py:29-36; beta is a zero buffer)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        M, D = x.shape
        y = _f32((M, D), x.device)
        L.layernorm(x, gamma, beta, M, D, out2=y, eps=eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        M, D = x.shape
        dx = _f32((M, D), x.device)
        dg, _ = L.layernorm_bwd(x, gamma.detach(), dy.contiguous(), dx, M, D, eps=ctx.eps)
        return dx, dg, None, None


class _FFBlock(torch.autograd.Function):
    """x + Linear(inner, dim)(GEGLU(Linear(dim, 2 inner)(nn.LayerNorm(x))))   (attention.py:45-52 + the residual of :330).
    img: the block's entry of `WeightImages.feedforward` (None: the four weight images are packed here, per call)."""

    @staticmethod
    def _images(w1, w2, dtype, dev):
        wi = WeightImages(dtype, dev)
        img = wi.feedforward(w1, w2)
        wi.refresh()
        return img

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, w2, dtype, eps, img):
        M, D = x.shape
        F = w2.shape[1]
        Fp = round_up(F, 8)
        dev = x.device
        if img is None:
            img = _FFBlock._images(w1, w2, dtype, dev)
        xn = _f32((M, D), dev)
        xa = torch.empty((M, D), device=dev, dtype=torch.bfloat16) if dtype == L.BF16 else None    # bf16 mode: the A operand of the first Linear, from the same launch
        L.layernorm(x, ln_w, ln_b, M, D, out=xa, out2=xn, eps=eps)
        # value rows [0, F) and gate rows [Fp, Fp + F) of the K-padded 2 Fp-row weight image; the pad rows stay zero -> h pad columns are 0
        h = _f32((M, 2 * Fp), dev)
        L.gemm(dtype, xa if xa is not None else xn, img['w1p'], M, 2 * Fp, D, C=h)
        a = _f32((M, Fp), dev)
        L.geglu(h, Fp, a, M, Fp)
        y = _f32((M, D), dev)
        L.gemm(dtype, a_operand(dtype, a), img['w2'], M, D, Fp, C=y, res=x)
        ctx.save_for_backward(x, ln_w, w1, w2, xn, h, a)
        ctx.dtype, ctx.eps, ctx.Fp, ctx.img = dtype, eps, Fp, img
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ln_w, w1, w2, xn, h, a = ctx.saved_tensors
        dtype, Fp, img = ctx.dtype, ctx.Fp, ctx.img
        M, D = x.shape
        F = w2.shape[1]
        dev = x.device
        dy = dy.contiguous()
        q = _q(dtype)
        Mp = round_up(M, q)
        sums = []
        # ---- second Linear: da = dy W2 (pad columns: zero rows of the W2^T image), dW2 = dy^T a
        dyT, aT, dyA = transposes(dtype, [(dy, 'a'), (a, 'w')], [dy])     # (D, Mp), (Fp, Mp)
        da = _f32((M, Fp), dev)
        L.gemm(dtype, dyA, img['w2t'], M, Fp, D, C=da)
        dW2 = None if F % 4 else _f32((D, F), dev)
        dW2p = None
        if F % 4:                                                        # inner 1365: the product on the padded width (a's pad columns are zero), then the slice
            dW2p = _f32((D, Fp), dev)
            _weight_grad_gemm(dtype, dyT, aT, D, Fp, Mp, dW2p, sums)
        else:
            _weight_grad_gemm(dtype, dyT, aT, D, F, Mp, dW2, sums)
        # ---- GEGLU
        dh = _f32((M, 2 * Fp), dev)
        L.geglu_bwd(h, Fp, da, dh, M, Fp)
        # ---- first Linear: dxn = dh W1 (the padded layout, transposed), dW1 = dh^T xn in two row groups (value | gate)
        dhT, xnT, dhA = transposes(dtype, [(dh, 'a'), (xn, 'w')], [dh])   # (2 Fp, Mp), (D, Mp)
        dxn = _f32((M, D), dev)
        L.gemm(dtype, dhA, img['w1t'], M, D, 2 * Fp, C=dxn)
        dW1 = _f32((2 * F, D), dev)
        _weight_grad_gemm(dtype, dhT[:F], xnT, F, D, Mp, dW1[:F], sums)
        _weight_grad_gemm(dtype, dhT[Fp:Fp + F], xnT, F, D, Mp, dW1[F:], sums)
        # ---- LayerNorm + residual; the K-slice sums of the three weight gradients and the dgamma | dbeta column sum leave in one launch
        cs = []
        dx = _f32((M, D), dev)
        dg, db = L.layernorm_bwd(x, ln_w.detach(), dxn, dx, M, D, add=dy, want_beta=True, eps=ctx.eps, defer=cs)
        L.reduce_multi(sums, cs)
        if dW2p is not None:
            dW2 = dW2p[:, :F].contiguous()
        return dx, dg, db, dW1, dW2, None, None, None


class _PEGBlock(torch.autograd.Function):
    """x + dsconv(pad(x)) + b   (attention.py:57-85, :323)"""

    @staticmethod
    def forward(ctx, x, weight, bias, shape, causal):
        b, t, h, w = shape
        M, D = x.shape
        taps = weight.detach().reshape(D, 27).t().contiguous()            # (27, D): the layout pk_peg reads (27 x 512 floats)
        y = _f32((M, D), x.device)
        L.peg(x, taps, bias.detach(), y, b, t, h, w, D, causal)
        ctx.save_for_backward(x, taps)
        ctx.shape, ctx.causal, ctx.wshape = shape, causal, weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, taps = ctx.saved_tensors
        b, t, h, w = ctx.shape
        M, D = x.shape
        dy = dy.contiguous()
        dx = _f32((M, D), x.device)
        cs = []                                                          # the tap partials and the bias gradient: one column-sum launch
        dtaps = L.peg_bwd(dy, x, taps, dx, b, t, h, w, D, ctx.causal, defer=cs)
        dbias = L.colsum_deferred(dy, M, D, _f32((D,), x.device), cs)
        if cs:
            L.colsum_multi(cs)
        return dx, dtaps.t().contiguous().reshape(ctx.wshape), dbias, None, None


class _AttnBlock(torch.autograd.Function):
    """x + to_out(softmax(q^ k^T + bias) v)   (attention.py:89-182 + the residuals of :325-328); self-attention reads K / V from the
    UN-normalised x (:140-144), cross-attention from context_norm(context)."""

    @staticmethod
    def forward(ctx, x, context, gamma, beta, cgamma, cbeta, wq, wkv, null_kv, q_scale, k_scale, wo, bias, kmask, meta):
        dtype, S, n, n_ctx, heads, scale, eps, slopes, img = meta
        if img is None:                                                  # a direct caller (no WeightImages of the whole Transformer): pack here
            wi = WeightImages(dtype, x.device)
            img = dict(wq=wi.both(wq), wkv=wi.both(wkv), wo=wi.both(wo))
            wi.refresh()
            meta = meta[:-1] + (img,)
        dev = x.device
        M, D = x.shape
        inner = wq.shape[0]
        nnull = null_kv.shape[1] // 2
        is_cross = context is not None
        n_kv = n_ctx if is_cross else n
        xn = _f32((M, D), dev)
        # bf16 mode: the A operands of to_q (LayerNorm output) and of the self-attention's to_kv (the un-normalised x) as bf16 copies from the same launch
        xa = torch.empty((M, D), device=dev, dtype=torch.bfloat16) if dtype == L.BF16 else None
        xr = torch.empty((M, D), device=dev, dtype=torch.bfloat16) if (dtype == L.BF16 and context is None) else None
        L.layernorm(x, gamma, beta, M, D, out=xa, out2=xn, raw=xr, eps=eps)
        if is_cross:
            Mk, Dk = context.shape
            if cgamma is not None:
                src = _f32((Mk, Dk), dev)
                L.layernorm(context, cgamma, cbeta, Mk, Dk, out2=src, eps=eps)
            else:
                src = context
        else:
            src = x
        q = linear_fwd(dtype, xa if xa is not None else xn, wq, Wimg=img['wq'][0])
        kv = linear_fwd(dtype, xr if xr is not None else src, wkv, Wimg=img['wkv'][0])
        td = L.tdtype(dtype)
        nq_pad, nk_pad = L.attn_pads(n, n_kv, nnull)
        Qp = torch.empty((S * heads * nq_pad * 64,), device=dev, dtype=td)
        Kp = torch.empty((S * heads * nk_pad * 64,), device=dev, dtype=td)
        Vt = torch.empty((S * heads * nk_pad * 64,), device=dev, dtype=td)
        L.attn_prep(dtype, q, kv, null_kv.detach(), q_scale.detach(), k_scale.detach(), float(scale), Qp, Kp, Vt, S, heads, n, n_kv, nnull)
        o = _f32((M, inner), dev)
        # every score row's log-sum-exp: the backward kernels start from it.  P = exp(s - lse) must use the lse of the SAME scores: in the bf16 mode the
        # forward's scores are products of bf16 operands, so the hand-over needs the backward's single-bf16-product form (ATTN_BWD_BF16); with the
        # split-bf16 backward (PK_ATTN_BWD_BF16=0) that mode keeps the extra pass
        lse = _f32((S * heads * n,), dev) if (dtype != L.BF16 or ATTN_BWD_BF16) else None
        L.attn_fwd(dtype, Qp, Kp, Vt, o, S, heads, n, n_kv, nnull, bias=bias, kmask=kmask, slopes=slopes, causal=slopes is not None, lse=lse)
        y = _f32((M, D), dev)
        L.gemm(dtype, a_operand(dtype, o), img['wo'][0], M, D, inner, C=y, res=x)
        ctx.save_for_backward(x, context, gamma, cgamma, wq, wkv, null_kv, q_scale, k_scale, wo, bias, kmask, xn, src, q, kv, o, lse)
        ctx.meta = meta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, context, gamma, cgamma, wq, wkv, null_kv, q_scale, k_scale, wo, bias, kmask, xn, src, q, kv, o, lse = ctx.saved_tensors
        dtype, S, n, n_ctx, heads, scale, eps, slopes, img = ctx.meta
        dev = x.device
        M, D = x.shape
        inner = wq.shape[0]
        nnull = null_kv.shape[1] // 2
        is_cross = context is not None
        n_kv = n_ctx if is_cross else n
        nkt = nnull + n_kv
        dy = dy.contiguous()
        sums = []
        # ---- to_out
        dyT, oT, dyA = transposes(dtype, [(dy, 'a'), (o, 'w')], [dy])
        do, dWo = linear_bwd(dtype, o, wo, dy, Wt=img['wo'][1], dyT=dyT, xT=oT, defer=sums, dyA=dyA)
        # ---- attention core
        Qh, Kh, Vh = _f32((S * heads * n, 64), dev), _f32((S * heads * nkt, 64), dev), _f32((S * heads * nkt, 64), dev)
        L.attn_train_prep(q, kv, null_kv.detach(), q_scale.detach(), k_scale.detach(), float(scale), Qh, Kh, Vh, S, heads, n, n_kv, nnull)
        dQh, dKh, dVh = torch.empty_like(Qh), torch.empty_like(Kh), torch.empty_like(Vh)
        want_dbias = bias is not None and ctx.needs_input_grad[12]
        dS = _f32((S, heads * n * n_kv), dev) if want_dbias else None
        L.attn_bwd(Qh, Kh, Vh, o, do, dQh, dKh, dVh, S, heads, n, n_kv, nnull, bias=bias, kmask=kmask, dS=dS, slopes=slopes, causal=slopes is not None,
                   split_bf16=dtype != L.F32, lse=lse, bf16_products=(dtype == L.BF16 and ATTN_BWD_BF16))
        dbias = None
        if want_dbias:
            dbias = _f32(tuple(bias.shape), dev)
            L.sum_batch(dS, S, dbias, heads * n * n_kv)
        dq, dkv = _f32((M, inner), dev), _f32(tuple(kv.shape), dev)
        cs = []                                                          # dq_scale | dk_scale and the LayerNorm gains: one column-sum launch at the end
        dqs, dks, dnull = L.attn_train_prep_bwd(q, kv, null_kv.detach(), q_scale.detach(), k_scale.detach(), float(scale), dQh, dKh, dVh, dq, dkv,
                                                S, heads, n, n_kv, nnull, defer=cs)
        if dnull is None:
            dnull = torch.zeros_like(null_kv)
        # ---- projections, LayerNorms, residual
        dqT, xnT, dkvT, srcT, dqA, dkvA = transposes(dtype, [(dq, 'a'), (xn, 'w'), (dkv, 'a'), (src, 'w')], [dq, dkv])
        dxn, dWq = linear_bwd(dtype, xn, wq, dq, Wt=img['wq'][1], dyT=dqT, xT=xnT, defer=sums, dyA=dqA)
        dx = _f32((M, D), dev)
        dcg = dctx = None
        if is_cross:
            need_ctx = ctx.needs_input_grad[1]
            dsrc, dWkv = linear_bwd(dtype, src, wkv, dkv, need_dx=(cgamma is not None) or need_ctx, Wt=img['wkv'][1], dyT=dkvT, xT=srcT, defer=sums, dyA=dkvA)
            if cgamma is not None:
                dctx = _f32(tuple(context.shape), dev)
                dcg, _ = L.layernorm_bwd(context, cgamma.detach(), dsrc, dctx, context.shape[0], context.shape[1], eps=eps, defer=cs)
            else:
                dctx = dsrc
            dg, _ = L.layernorm_bwd(x, gamma.detach(), dxn, dx, M, D, add=dy, eps=eps, defer=cs)
        else:
            t, dWkv = linear_bwd(dtype, x, wkv, dkv, add=dy, Wt=img['wkv'][1], dyT=dkvT, xT=srcT, defer=sums, dyA=dkvA)   # dy + dkv Wkv: K / V read the un-normalised x
            dg, _ = L.layernorm_bwd(x, gamma.detach(), dxn, dx, M, D, add=t, eps=eps, defer=cs)
        L.reduce_multi(sums, cs)                                         # the K-slice sums of dWo / dWq / dWkv and the column sums: one launch
        return dx, dctx, dg, None, dcg, None, dWq, dWkv, dnull, dqs, dks, dWo, dbias, None, None


class _Embed(torch.autograd.Function):
    """tok[ids] + pos[arange(n)], gradient scaled by alpha (phenaki_pytorch.py:194-199; TokenCritic: alpha = 1)"""

    @staticmethod
    def forward(ctx, tok, pos, ids, alpha):
        b, n = ids.shape
        D = tok.shape[1]
        x = _f32((b * n, D), tok.device)
        L.embed(ids, tok.detach(), pos.detach(), x, b, n, D)
        ctx.save_for_backward(ids)
        ctx.alpha, ctx.shapes = alpha, (tuple(tok.shape), tuple(pos.shape))
        return x

    @staticmethod
    def backward(ctx, dy):
        ids, = ctx.saved_tensors
        b, n = ids.shape
        (V1, D), (P, _) = ctx.shapes
        dy = dy.contiguous()
        dtok, dpos = _zeros((V1, D), dy.device), _zeros((P, D), dy.device)
        L.embed_bwd(dy, ids, ctx.alpha, dtok, dpos, b, n, D)
        return dtok, dpos, None, None


_REL_TABLES = {}


def _rel_table(dims, device):
    """cached per (grid, device): see _build_rel_table"""
    key = (tuple(dims), str(device))
    if key not in _REL_TABLES:
        _REL_TABLES[key] = _build_rel_table(dims, device)
    return _REL_TABLES[key]


def _build_rel_table(dims, device):
    """relative-position table of a grid (attention.py:257-268): features (Lt, 8) f32 = sign(d) log(|d| + 1) per axis (zero-padded to 8
    columns), the mixed-radix position code (n,) int32 and its offset: bias[h][i][j] = mlp(features)[code[i] - code[j] + off][h]"""
    strides, acc = [], 1
    for d in reversed(dims):
        strides.insert(0, acc)
        acc *= 2 * d - 1
    grids = torch.meshgrid(*[torch.arange(d) for d in dims], indexing='ij')
    code = sum(g.reshape(-1) * st for g, st in zip(grids, strides))
    off = sum((d - 1) * st for d, st in zip(dims, strides))
    l = torch.arange(acc)
    feats = torch.zeros((acc, 8))
    for a, (d, st) in enumerate(zip(dims, strides)):
        delta = (l // st) % (2 * d - 1) - (d - 1)
        feats[:, a] = torch.sign(delta) * torch.log(delta.abs().float() + 1)
    return feats.to(device), code.to(torch.int32).to(device), int(off)


class _PositionBias(torch.autograd.Function):
    """ContinuousPositionBias (attention.py:229-275) evaluated on the prod(2 d - 1) distinct relative positions instead of all n^2 pairs
    (same function of the same inputs), gathered to (heads, n, n); exact f32 like the reference (rel_pos.float()).  The first Linear's
    num_dims input columns are zero-padded to the 8 feature columns, the last Linear's `heads` outputs to a multiple of 4 (zero rows)."""

    @staticmethod
    def _padded(w, b, li, nl, dev):
        """(weight, bias) of layer li as the GEMM sees them"""
        N, K = w.shape
        Np = round_up(N, 4) if li == nl - 1 else N
        Kp = 8 if li == 0 else K
        if (Np, Kp) == (N, K):
            return w.detach(), b.detach()
        wp, bp = _zeros((Np, Kp), dev), _zeros((1, Np), dev)
        L.pack(w.detach(), N, K, False, wp, Kp, 0)
        L.pack(b.detach().view(1, N), 1, N, False, bp, Np, 0)
        return wp, bp.view(Np)

    @staticmethod
    def forward(ctx, feats, code, off, n, *params):
        dev = feats.device
        ws, bs = params[0::2], params[1::2]
        nl = len(ws)
        acts = [feats]
        hcur = feats
        for li, (w, b) in enumerate(zip(ws, bs)):
            wp, bp = _PositionBias._padded(w, b, li, nl, dev)
            hcur = linear_fwd(L.F32, hcur, wp, bias=bp, act=L.ACT_NONE if li == nl - 1 else L.ACT_LEAKY)
            acts.append(hcur)
        heads = ws[-1].shape[0]
        out = _f32((heads, n, n), dev)
        L.bias_gather(hcur, code, off, out, heads, n)
        ctx.save_for_backward(code, *params, *acts)
        ctx.off, ctx.n, ctx.nl = off, n, nl
        return out

    @staticmethod
    def backward(ctx, dbias):
        nl = ctx.nl
        code = ctx.saved_tensors[0]
        params = ctx.saved_tensors[1:1 + 2 * nl]
        acts = ctx.saved_tensors[1 + 2 * nl:]
        ws, bs = params[0::2], params[1::2]
        dev = code.device
        heads = ws[-1].shape[0]
        g = _zeros(tuple(acts[-1].shape), dev)                          # (Lt, heads padded to 4): the pad columns stay zero
        L.bias_scatter(dbias.contiguous(), code, ctx.off, g, heads, ctx.n)
        grads = [None] * (2 * nl)
        for li in range(nl - 1, -1, -1):
            w, xin, yout = ws[li], acts[li], acts[li + 1]
            N, K = w.shape
            if li != nl - 1:                                            # LeakyReLU(0.1) behind every layer but the last
                dz = _f32(tuple(yout.shape), dev)
                L.leaky_bwd(yout, g, dz, yout.shape[0], yout.shape[1], 0.1)
                g = dz
            grads[2 * li + 1] = L.colsum(g, g.shape[0], g.shape[1], _f32((g.shape[1],), dev))[:N]
            wp, _ = _PositionBias._padded(w, bs[li], li, nl, dev)
            g_in, dwp = linear_bwd(L.F32, xin, wp, g, need_dx=li > 0)
            grads[2 * li] = dwp[:N, :K].contiguous()
            g = g_in
        return (None, None, None, None, *grads)


class _VocabCrossEntropy(torch.autograd.Function):
    """mean over the selected rows of CE(E[rows] W^T + b, targets[rows]); the (rows, V) logits are never stored (phenaki_pytorch.py:640-643)"""

    @staticmethod
    def forward(ctx, embeds, weight, bias, targets, rows, dtype, slab, imgs=None, shared=None):
        """imgs: (image of W, image of W^T) from `linear_images` (already refreshed), else packed here.  shared: (partials, M_all) of a
        pk_vocab_sample call with need_lse over ALL rows of `embeds` on the same weight image (the critic's gumbel sampling of the training step,
        phenaki_pytorch.py:653-655): its per-tile (max, sum-exp) statistics are temperature- and noise-free, so the rows of this loss take theirs
        from it instead of a second pass over the vocabulary."""
        L.require_device(embeds, 'embeds')
        R, D = embeds.shape
        V = weight.shape[0]
        assert V % 8 == 0 and D % 8 == 0, 'vocab_cross_entropy: vocabulary size and dim must be multiples of 8 (16-byte operand rows)'
        dev = embeds.device
        E = embeds.detach().float().contiguous()
        M = R if rows is None else rows.numel()
        # the rows the head works on, gathered once: the operand of the forward products and of both gradient products
        td = L.tdtype(dtype)
        if rows is None:
            A = E.to(td) if td != torch.float32 else E
        else:
            A = torch.empty((M, D), device=dev, dtype=td)
            L.pack(E, M, D, False, A, D, 1 if td == torch.bfloat16 else 0, rows=rows)
        Wp = imgs[0] if imgs is not None else _operand(weight.detach().float(), dtype)
        b = bias.detach().float().contiguous() if bias is not None else torch.zeros((V,), device=dev)
        tg = targets.detach().long().contiguous()
        nt = L.vocab_ntiles(V)
        partials = torch.empty((5 * nt * M,), device=dev, dtype=torch.float32)
        if shared is not None:
            pa, M_all = shared
            stats = pa.view(5, nt, M_all)[3:5]                            # (max | sum-exp) planes, [tile][row]
            dst = partials.view(5, nt, M)[3:5]
            if rows is None:
                dst.copy_(stats)
            else:
                torch.index_select(stats, 2, rows.long(), out=dst)
        else:
            L.vocab_sample(dtype, A, Wp, b, M, V, D, 1.0, None, None, 0, True, partials, no_noise=True)
        loss_rows = torch.empty((M,), device=dev, dtype=torch.float32)
        lse = torch.empty((M,), device=dev, dtype=torch.float32)
        L.vocab_ce(dtype, partials, M, V, A, Wp, b, D, tg, rows, loss_rows, lse=lse)
        # A (the gathered operand rows) and Wp (the packed vocabulary weight: 64-134 MB at V = 65 536) go through save_for_backward like the rest,
        # so they are released when autograd frees the saved tensors and take part in its version checks
        ctx.save_for_backward(E, weight, b, tg, lse, rows, A, Wp)
        ctx.dtype, ctx.slab, ctx.has_bias, ctx.Wt = dtype, slab, bias is not None, (imgs[1] if imgs is not None else None)
        return L.colsum(loss_rows.view(M, 1), M, 1, torch.empty((1,), device=dev, dtype=torch.float32), scale=1.0 / M).reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        E, weight, b, tg, lse, rows, A, Wp = ctx.saved_tensors
        dtype, slab = ctx.dtype, ctx.slab
        R, D = E.shape
        M = A.shape[0]
        V = weight.shape[0]
        dev = E.device
        td = L.tdtype(dtype)
        q = _q(dtype)
        Mp = round_up(M, q)
        scale = 1.0 / M
        gdev = grad_out.detach().float().reshape(1).contiguous()         # the upstream gradient is read on the device: no host sync in backward
        # operands of the two gradient products (W-side images: rows = output features, K along the contraction)
        Wt = ctx.Wt if ctx.Wt is not None else pack_operand(weight.detach(), dtype, transpose=True)   # (D, Vp): dE = g @ W   -> "W" operand = W^T, K = vocabulary
        Af = A.float() if A.dtype != torch.float32 else A
        Et = pack_operand(Af, dtype, transpose=True)                             # (D, Mp): dW = g^T @ E -> "W" operand = E^T, K = rows
        dE = torch.empty((M, D), device=dev, dtype=torch.float32)
        dW = torch.empty((V, D), device=dev, dtype=torch.float32)
        db = torch.empty((V,), device=dev, dtype=torch.float32)
        Vs_max = min(slab, V)
        logits = torch.empty((M, Vs_max), device=dev, dtype=torch.float32)
        g = torch.empty((M, Vs_max), device=dev, dtype=td)
        gT = torch.empty((Vs_max, Mp), device=dev, dtype=td)
        first = True
        for v0 in range(0, V, slab):
            Vs = min(slab, V - v0)
            # logits of the slab: the same A / W rows / bias as the forward pass (rows [v0, v0 + Vs) of the packed weight)
            L.gemm(dtype, A, Wp[v0:v0 + Vs], M, Vs, D, C=logits, bias=b[v0:v0 + Vs], ldc=logits.stride(0))
            L.ce_grad_slab(logits, lse, tg, rows, M, Vs, v0, scale, g, gT, db=db[v0:v0 + Vs], scale_dev=gdev)
            # dE (+)= g @ W_slab: contraction over the slab's columns = columns [v0, v0 + Vs) of the W^T image (k offset on the operand)
            L.gemm(dtype, g, Wt[:, v0:], M, D, Vs, C=dE, res=None if first else dE, lda=g.stride(0))
            # dW_slab = g^T @ E (K = the rows, padded to the k-tile: gT's pad columns are zeroed by the kernel, E^T is zero-padded)
            _weight_grad_gemm(dtype, gT[:Vs], Et, Vs, D, Mp, dW[v0:v0 + Vs])
            first = False
        if rows is not None:
            full = _zeros((R, D), dev)
            L.scatter_rows(dE, rows, full, M, D)
            dE = full
        return dE, dW, (db if ctx.has_bias else None), None, None, None, None, None, None


class _Linear(torch.autograd.Function):
    """y = x W^T + b on (M, K) f32 rows (MaskGit.to_logits when a caller wants the logits themselves under autograd)"""

    @staticmethod
    def forward(ctx, x, W, b, dtype):
        ctx.save_for_backward(x, W)
        ctx.dtype, ctx.has_bias = dtype, b is not None
        return linear_fwd(dtype, x, W.detach(), bias=b.detach() if b is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        dx, dW = linear_bwd(ctx.dtype, x, W.detach(), dy, need_dx=ctx.needs_input_grad[0])
        db = L.colsum(dy, dy.shape[0], dy.shape[1], _f32((dy.shape[1],), dy.device)) if ctx.has_bias else None
        return dx, dW, db, None


class _RowDot(torch.autograd.Function):
    """z = e w^T + b for a single output unit (the critic heads: Linear(dim, 1) + Rearrange('... 1 -> ...'), phenaki_pytorch.py:246-249)"""

    @staticmethod
    def forward(ctx, e, w, b):
        M, D = e.shape
        z = _f32((M,), e.device)
        L.bce_head(e, w.detach().reshape(-1), b.detach(), None, M, D, logits=z)
        ctx.save_for_backward(e, w)
        return z

    @staticmethod
    def backward(ctx, dz):
        e, w = ctx.saved_tensors
        M, D = e.shape
        dev = e.device
        dzp = _f32((M, 4), dev)                                         # the one output unit padded to 4 columns for the GEMM's vector epilogue
        L.pack(dz.contiguous().view(M, 1), M, 1, False, dzp, 4, 0)
        wp = _zeros((4, D), dev)
        L.pack(w.detach().reshape(1, D), 1, D, False, wp, D, 0)
        de, dwp = linear_bwd(L.F32, e, wp, dzp)
        db = L.colsum(dzp, M, 1, _f32((1,), dev), ld=4)
        return de, dwp[:1].reshape(w.shape).contiguous(), db


class _BCEHead(torch.autograd.Function):
    """mean BCE-with-logits of (e w^T + b) against the labels (phenaki_pytorch.py:246-249 / :320-322 heads, :673-676 loss)"""

    @staticmethod
    def forward(ctx, e, w, b, labels):
        M, D = e.shape
        loss_rows = _f32((M, 1), e.device)
        L.bce_head(e, w.detach().reshape(-1), b.detach(), labels, M, D, loss_rows=loss_rows)
        ctx.save_for_backward(e, w, b, labels)
        return L.colsum(loss_rows, M, 1, _f32((1,), e.device), scale=1.0 / M).reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        e, w, b, labels = ctx.saved_tensors
        M, D = e.shape
        de = _f32((M, D), e.device)
        dw, db = L.bce_head(e, w.detach().reshape(-1), b.detach(), labels, M, D, scale=1.0 / M, de=de,
                            scale_dev=grad_out.detach().float().reshape(1).contiguous())
        return de, dw.reshape(w.shape), db.reshape(b.shape), None


# ------------------------------------------------------------------------------------------------------------ modules -> blocks

def vocab_cross_entropy(embeds, weight, bias, targets, compute_dtype='bf16x3', slab=2048, rows=None):
    """mean_m CE(embeds[r_m] @ weight^T + bias, targets[r_m]) with gradients for embeds / weight / bias; logits never stored.
    embeds (R, D) f32 on the HIP device, weight (V, D), bias (V,) or None, targets (R,) int64 in [0, V); rows (M,) int32: the rows r_m the
    loss is taken over (None: all).  slab: vocabulary columns per backward step (a multiple of 64)."""
    assert slab % 64 == 0 and slab > 0
    assert weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0, 'vocabulary size and embedding width must be multiples of 8'
    return _VocabCrossEntropy.apply(embeds, weight, bias, targets, rows, resolve_dtype(compute_dtype), int(slab), None, None)


def layernorm_train(ln: LayerNorm, x2d):
    return _LayerNormFn.apply(x2d, ln.gamma, ln.beta, ln.eps)


def feedforward_train(ff: FeedForwardSeq, x2d, dtype, img=None):
    ln, lin1, lin2 = ff[0], ff[1], ff[4]
    assert ff[3].p == 0., 'the training kernels are built for ff_dropout = 0 (the reference default)'
    return _FFBlock.apply(x2d, ln.weight, ln.bias, lin1.weight, lin2.weight, dtype, ln.eps, img)


def peg_train(peg: PEG, x2d, shape):
    return _PEGBlock.apply(x2d, peg.dsconv.weight, peg.dsconv.bias, tuple(shape), peg.causal)


def attention_train(attn: Attention, x2d, S, n, dtype, *, context2d=None, n_ctx=None, attn_bias=None, kmask=None, img=None):
    assert attn.attn_dropout.p == 0., 'the training kernels are built for attn_dropout = 0 (the reference default)'
    assert not (attn.causal and context2d is not None), 'causal attention is self-attention (attention.py:166-172)'
    cn = attn.context_norm if isinstance(attn.context_norm, LayerNorm) else None
    slopes = attn.rel_pos_bias.slopes.reshape(-1).contiguous() if attn.causal else None     # ALiBi + causal mask inside the kernels
    meta = (dtype, S, n, n_ctx, attn.heads, float(attn.scale), attn.norm.eps, slopes, img)
    return _AttnBlock.apply(x2d, context2d, attn.norm.gamma, attn.norm.beta, cn.gamma if (cn is not None and context2d is not None) else None,
                            cn.beta if cn is not None else None, attn.to_q.weight, attn.to_kv.weight, attn.null_kv, attn.q_scale, attn.k_scale,
                            attn.to_out.weight, attn_bias, kmask, meta)


def position_bias_train(cpb, dims, device):
    feats, code, off = _rel_table(tuple(dims), device)
    n = 1
    for d in dims:
        n *= d
    params = []
    for layer in cpb.net:
        lin = layer[0] if isinstance(layer, torch.nn.Sequential) else layer
        params += [lin.weight, lin.bias]
    return _PositionBias.apply(feats, code, off, n, *params)


def transformer_train(tr, x2d, S, n, dtype, *, video_shape=None, attn_bias=None, context2d=None, n_ctx=None, self_attn_mask=None,
                      cross_attn_context_mask=None):
    """attention.py:315-332 on (S n, D) f32 rows, every block an autograd Function"""
    x = x2d
    wi = transformer_images(tr, dtype)
    wi.refresh()                                                         # ONE launch: every projection weight of the stack, W and W^T images
    for (peg, self_attn, cross_attn, ff), img in zip(tr.layers, wi.layers):
        if exists(peg):
            x = peg_train(peg, x, video_shape)
        x = attention_train(self_attn, x, S, n, dtype, attn_bias=attn_bias, kmask=self_attn_mask, img=img['sa'])
        if exists(cross_attn) and exists(context2d):
            x = attention_train(cross_attn, x, S, n, dtype, context2d=context2d, n_ctx=n_ctx, kmask=cross_attn_context_mask, img=img['ca'])
        x = feedforward_train(ff, x, dtype, img['ff'])
    return layernorm_train(tr.norm_out, x)


def _u8(mask):
    return None if mask is None else mask.to(torch.uint8).contiguous()


def trunk_train(model, ids2d, video_patch_shape, *, context=None, text_mask=None, video_mask=None, use_bias=False, use_cross=True, alpha=1.0):
    """MaskGit.forward(return_embeds=True) / the TokenCritic trunk with gradients: (b n, D) f32 rows of norm_out"""
    b, n = ids2d.shape
    dt = compute_dtype_of(model)
    x = _Embed.apply(model.token_emb.weight, model.pos_emb.weight, ids2d.long().contiguous(), float(alpha))
    bias = position_bias_train(model.continuous_pos_bias, video_patch_shape, x.device) if use_bias else None
    ctx2, n_ctx = None, None
    if use_cross and exists(context):
        n_ctx = context.shape[1]
        ctx2 = context.reshape(b * n_ctx, context.shape[-1]).float().contiguous()
    return transformer_train(model.transformer, x, b, n, dt, video_shape=(b, *video_patch_shape), attn_bias=bias, context2d=ctx2, n_ctx=n_ctx,
                             self_attn_mask=_u8(video_mask), cross_attn_context_mask=_u8(text_mask) if ctx2 is not None else None)


def wants_grad(*modules):
    """grad mode on and some parameter of the given modules is trainable: the call has to build an autograd graph"""
    return torch.is_grad_enabled() and any(p.requires_grad for m in modules if m is not None for p in m.parameters())


def maskgit_forward_train(mg, x, *, cond_drop_prob=0., text_mask=None, video_mask=None, video_patch_shape=None, return_embeds=False, context=None):
    """MaskGit.forward (phenaki_pytorch.py:163-213) with an autograd graph: logits (b, n, num_tokens) or, return_embeds, the trunk output"""
    from .phenaki import prob_mask_like
    x, vps = mg._prepare(x, text_mask, video_patch_shape)
    b, n = x.shape
    if exists(context) and not exists(text_mask):
        text_mask = torch.ones(context.shape[:2], device=x.device, dtype=torch.bool)
    if cond_drop_prob > 0 and exists(text_mask):
        keep_mask = prob_mask_like((b,), 1 - cond_drop_prob, device=x.device)
        text_mask = keep_mask[:, None] & text_mask
    e = trunk_train(mg, x, vps, context=context, text_mask=text_mask, video_mask=video_mask, use_bias=True, use_cross=not mg.unconditional,
                    alpha=mg.gradient_shrink_alpha)
    if return_embeds:
        return e.view(b, n, mg.dim)
    logits = _Linear.apply(e, mg.to_logits.weight, mg.to_logits.bias, compute_dtype_of(mg))
    return logits.view(b, n, -1)


def critic_forward_train(critic, x, *, text_mask=None, cond_drop_prob=None, context=None, video_mask=None, video_patch_shape=None):
    """TokenCritic.forward (phenaki_pytorch.py:265-302) / SelfCritic.forward (:334-336) with an autograd graph: scores (b, n)"""
    from .phenaki import SelfCritic, prob_mask_like
    if isinstance(critic, SelfCritic):
        e = maskgit_forward_train(critic.maskgit, x, cond_drop_prob=cond_drop_prob or 0., text_mask=text_mask, video_mask=video_mask,
                                  video_patch_shape=video_patch_shape, return_embeds=True, context=context)
        b, n, D = e.shape
        head = critic.to_pred[0]
        return _RowDot.apply(e.reshape(b * n, D), head.weight, head.bias).view(b, n)
    x, vps = critic._flatten(x, video_patch_shape)
    b, n = x.shape
    if exists(context) and not exists(text_mask):
        text_mask = torch.ones(context.shape[:2], device=x.device, dtype=torch.bool)
    if exists(context) and exists(cond_drop_prob) and cond_drop_prob > 0:
        keep_mask = prob_mask_like((b,), 1 - cond_drop_prob, device=x.device)
        text_mask = keep_mask[:, None] & text_mask
    e = trunk_train(critic, x, vps, context=context if critic.has_cross_attn else None, text_mask=text_mask, video_mask=video_mask,
                    use_bias=False, use_cross=critic.has_cross_attn, alpha=1.0)
    head = critic.to_logits[0]
    return _RowDot.apply(e, head.weight, head.bias).view(b, n)


def phenaki_loss(ph, videos=None, *, texts=None, video_codebook_ids=None, video_frame_mask=None, text_embeds=None, cond_drop_prob=None,
                 only_train_generator=False, only_train_critic=False, _draws=None):
    """phenaki_pytorch.py:562-687 with an autograd graph over the MaskGit / critic parameters (the C-ViViT and the T5 encoder are frozen there
    too).  _draws (tests): dict(rand_step, perm_noise, gumbel_u) replaces the three random draws, as in `Phenaki.objective_value`.
    `cond_drop_prob` is accepted for signature compatibility and HAS NO EFFECT: the reference overwrites it with 0 before use
    (phenaki_pytorch.py:594 shadows the argument), so classifier-free-guidance dropout never fires in its training step either."""
    from .phenaki import SelfCritic, TokenCritic
    assert not (only_train_generator and only_train_critic)
    assert not (only_train_critic and not exists(ph.critic)), 'only_train_critic needs a critic (Phenaki(critic=...) or self_token_critic=True)'
    assert exists(videos) ^ exists(video_codebook_ids), 'either raw video or video codebook ids must be given'
    assert not (exists(videos) and not exists(ph.cvivit)), 'cvivit must be provided if one wants to encode the videos live during training'
    assert (exists(text_embeds) ^ exists(texts)) ^ ph.unconditional, \
        'either raw text of text embeds must be given, and if unconditional, none should be given'
    assert not (exists(text_embeds) and text_embeds.shape[-1] != ph.text_embed_dim), 'text embedding dimension is not correct'
    mg, critic = ph.maskgit, ph.critic
    with torch.no_grad():
        if not exists(video_codebook_ids):
            assert videos.ndim in {4, 5}
            if videos.ndim == 4:
                videos = videos.unsqueeze(2)
            video_codebook_ids = ph.cvivit(videos, return_only_codebook_ids=True)
        L.require_device(video_codebook_ids, 'video_codebook_ids')
        assert video_codebook_ids.ndim == 4, 'video codebook ids must be (batch, frames, height, width): MaskGit takes the patch shape from it'
        device = video_codebook_ids.device
        patch_shape = tuple(video_codebook_ids.shape[1:])
        text_mask = None
        if not ph.unconditional:
            if not exists(text_embeds):
                text_embeds = ph.encode_texts(texts, output_device=device)
            text_embeds = text_embeds.to(device).float()
            text_mask = torch.any(text_embeds != 0, dim=-1)
        video_mask = None
        if exists(video_frame_mask):
            video_mask = ph.cvivit.calculate_video_token_mask(videos, video_frame_mask=video_frame_mask)
        ids = video_codebook_ids.reshape(video_codebook_ids.shape[0], -1).long().contiguous()
        b, n = ids.shape
        draws = _draws or {}
        # the step index is drawn on the HOST (reference: on the device, :620): the number of masked tokens per sample -- hence the row
        # count of the vocabulary head -- is then known without reading anything back, and the host can run a whole step ahead of the GPU
        rand_step = draws['rand_step'].cpu() if 'rand_step' in draws else torch.randint(0, ph.steps, (b,))
        mask_token_prob = torch.cos(rand_step * math.pi * 0.5 / ph.steps)
        perm_noise = draws['perm_noise'].to(device) if 'perm_noise' in draws else torch.rand((b, n), device=device)
        if exists(video_mask):                                                          # token counts live on the device: one read-back
            vm = video_mask
            num_tokens = vm.sum(dim=-1)                                                 # get_mask_subset_with_prob (phenaki_pytorch.py:43-55)
            num_masked = (mask_token_prob.to(device) * num_tokens).round().clamp(min=1)
            perm = perm_noise.argsort(dim=-1) - (n - num_tokens)[:, None]
            perm = perm.masked_fill(perm < 0, n)
            mask_token_mask = perm < num_masked[:, None]
            rows = mask_token_mask.reshape(-1).nonzero().reshape(-1).int()
        else:
            num_masked_host = (mask_token_prob * n).round().clamp(min=1)
            M_rows = int(num_masked_host.sum())
            # (pinned + non_blocking: a pageable host-to-device copy would wait for the stream, i.e. for the previous step)
            num_masked_dev = num_masked_host.pin_memory().to(device, non_blocking=True)
            mask_token_mask = perm_noise.argsort(dim=-1) < num_masked_dev[:, None]
            # the flat positions of the masked tokens in ascending order, with a size the host already knows (no nonzero() read-back)
            rows = torch.argsort((~mask_token_mask).reshape(-1).to(torch.uint8), stable=True)[:M_rows].int()
        masked_input = torch.where(mask_token_mask, ph.mask_id, ids)

    dt = compute_dtype_of(mg)
    D, V = mg.dim, mg.to_logits.weight.shape[0]
    ctx_mg = torch.no_grad() if only_train_critic else torch.enable_grad()
    with ctx_mg:
        e = trunk_train(mg, masked_input, patch_shape, context=text_embeds, text_mask=text_mask, video_mask=video_mask, use_bias=True,
                        use_cross=not mg.unconditional, alpha=mg.gradient_shrink_alpha)
    loss = None
    need_critic = exists(critic) and not only_train_generator
    head = None
    if mg.to_logits.weight.dtype == torch.float32 and mg.to_logits.weight.is_contiguous():
        head = linear_images(mg.to_logits, dt)                           # W and W^T images of the vocabulary head, one launch for both
        head.refresh()
    partials = None
    M = b * n
    if need_critic:
        # the critic's input: gumbel-sampled predictions at every position (phenaki_pytorch.py:653-659), no gradient through the ids.  ONE pass over
        # the vocabulary serves both this argmax and the cross entropy above it in the reference (round 6): the pass keeps the noise-free
        # (max, sum-exp) statistics of every row next to the noisy argmax
        with torch.no_grad():
            ed = e.detach()
            A = ed.to(L.tdtype(dt)) if L.tdtype(dt) != torch.float32 else ed
            w_logits = head.pair[0] if head is not None else pack_operand(mg.to_logits.weight.detach(), dt)
            partials = torch.empty((5 * L.vocab_ntiles(V) * M,), device=device, dtype=torch.float32)
            U = draws['gumbel_u'].to(device).float().contiguous() if 'gumbel_u' in draws else None
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if U is None else 0
            share = (not only_train_critic) and head is not None and mg.to_logits.bias is not None
            L.vocab_sample(dt, A, w_logits, mg.to_logits.bias.detach(), M, V, D, float(ph.critic_train_sample_temperature), U, None, seed, share, partials)
    if not only_train_critic:
        loss = _VocabCrossEntropy.apply(e, mg.to_logits.weight, mg.to_logits.bias, ids.reshape(-1), rows, dt, CE_SLAB, head.pair if head is not None else None,
                                        (partials, M) if (need_critic and share) else None)
    if not need_critic:
        return loss
    with torch.no_grad():
        pred = torch.empty((M,), device=device, dtype=torch.long)
        L.vocab_reduce(partials, M, V, None, None, None, pred, None, False)
        pred = pred.view(b, n)
        critic_input = torch.where(mask_token_mask, pred, ids)
        labels = (ids != pred).float().reshape(-1).contiguous()
    if isinstance(critic, SelfCritic):
        ce = trunk_train(critic.maskgit, critic_input, patch_shape, context=text_embeds, text_mask=text_mask, video_mask=video_mask, use_bias=True,
                         use_cross=not critic.maskgit.unconditional, alpha=critic.maskgit.gradient_shrink_alpha)
        head = critic.to_pred[0]
    else:
        assert isinstance(critic, TokenCritic)
        ce = trunk_train(critic, critic_input, patch_shape, context=text_embeds if critic.has_cross_attn else None, text_mask=text_mask,
                         video_mask=video_mask, use_bias=False, use_cross=critic.has_cross_attn, alpha=1.0)
        head = critic.to_logits[0]
    critic_loss = _BCEHead.apply(ce, head.weight, head.bias, labels)
    if only_train_critic:
        return critic_loss
    return _Axpy.apply(loss, critic_loss, float(ph.critic_loss_weight))


class _Axpy(torch.autograd.Function):
    """a + w * b on 0-d device tensors without an ATen kernel chain in the graph (two scalar gradients)"""

    @staticmethod
    def forward(ctx, a, b, w):
        ctx.w = w
        return a + b * w

    @staticmethod
    def backward(ctx, g):
        return g, g * ctx.w, None
