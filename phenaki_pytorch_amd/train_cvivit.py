"""The tokenizer's own training step on the MI355X kernels: `CViViT.forward(video)` with gradients for
`use_vgg_and_gan = False` -- reference /root/reference/phenaki_pytorch/cvivit.py:518-627 under autograd (caller cvivit_trainer.py:241-249):

    video -> patch embeddings -> spatial / temporal (causal, ALiBi) encoder -> LFQ (straight-through) -> temporal / spatial decoder
          -> to_pixels -> F.mse_loss(video, recon)

    loss = cvivit(video)            # grad mode on + trainable parameters -> cvivit_loss_train() below
    loss.backward()                 # every C-ViViT parameter that is on the path gets .grad

The transformer blocks are the autograd Functions of train.py (attention with the causal / ALiBi scores of attention.py:166-172 inside the
forward and backward kernels, the PEG with causal frame padding); this file adds what sits around them:

    _PatchEmbedFn   Rearrange + nn.LayerNorm(P) + Linear(P, dim) + nn.LayerNorm(dim)   pk_patchify_ln, pk_gemm, pk_layernorm / pk_layernorm_bwd,
                    (cvivit.py:273-285; the video itself takes no gradient)            pk_pack + pk_gemm, pk_mul + pk_colsum
    _GatherRows     '(b t) (h w)' <-> '(b h w) t' token order                           pk_pack (row gather) both ways
    _MergeFrames / _SplitFrames   first-frame rows | remaining rows <-> (b, t, h w)     pk_scatter_rows / pk_pack
    _LFQFn          project_in -> sign -> project_out, gradient straight through        pk_lfq_encode, pk_sign, pk_lfq_decode / pk_pack + pk_gemm
    _PatchMSE       mean (to_pixels(tokens) - patches(video))^2 -- un-patchify is a     pk_patchify_ln (raw rows), pk_sqdiff_partials
                    bijection of the pixels, so the loss is taken in patch layout        / pk_scaled_diff

The frame `mask` of variable-length training (cvivit.py:585-589) zeroes the dropped frames on both sides of the difference (pk_patch_frame_mask).
Not built: the discriminator / VGG / adaptive-weight branch (cvivit.py:604-671: torchvision's pretrained VGG16 is not available offline).
"""
import torch

from . import _lib as L
from .attention import compute_dtype_of
from .train import _Linear, _f32, linear_bwd, linear_fwd, position_bias_train, transformer_train

_INDEX_CACHE = {}


def _frame_indices(b, t, hw, device):
    """int32 row maps between the token orders of one video batch (cached): to_temporal[r'] = the '(b t) (h w)' row that lands on
    '(b h w) t' row r', to_spatial = its inverse, first / rest = the '(b t) (h w)' rows of frame 0 / frames 1.. in (b, [t - 1,] h w) order"""
    key = (b, t, hw, str(device))
    if key not in _INDEX_CACHE:
        bt = torch.arange(b * t * hw, dtype=torch.int32).view(b, t, hw)
        to_temporal = bt.permute(0, 2, 1).reshape(-1)
        to_spatial = torch.arange(b * hw * t, dtype=torch.int32).view(b, hw, t).permute(0, 2, 1).reshape(-1)
        first = bt[:, 0].reshape(-1)
        rest = bt[:, 1:].reshape(-1)
        _INDEX_CACHE[key] = tuple(x.contiguous().to(device) for x in (to_temporal, to_spatial, first, rest))
    return _INDEX_CACHE[key]


class _GatherRows(torch.autograd.Function):
    """y[r] = x[idx[r]] for a permutation idx with inverse inv"""

    @staticmethod
    def forward(ctx, x, idx, inv):
        M, D = x.shape
        ctx.save_for_backward(idx, inv)
        return L.pack(x, M, D, False, _f32((M, D), x.device), D, 0, rows=idx)

    @staticmethod
    def backward(ctx, dy):
        idx, inv = ctx.saved_tensors
        M, D = dy.shape
        return L.pack(dy.contiguous(), M, D, False, _f32((M, D), dy.device), D, 0, rows=inv), None, None


class _MergeFrames(torch.autograd.Function):
    """torch.cat((first_frame_tokens, rest_frames_tokens), dim = 1) on 2-D rows (cvivit.py:549)"""

    @staticmethod
    def forward(ctx, first, rest, idx_first, idx_rest):
        D = first.shape[1]
        M = first.shape[0] + rest.shape[0]
        x = _f32((M, D), first.device)
        L.scatter_rows(first, idx_first, x, first.shape[0], D)
        L.scatter_rows(rest, idx_rest, x, rest.shape[0], D)
        ctx.save_for_backward(idx_first, idx_rest)
        return x

    @staticmethod
    def backward(ctx, dx):
        idx_first, idx_rest = ctx.saved_tensors
        dx = dx.contiguous()
        D = dx.shape[1]
        nf, nr = idx_first.numel(), idx_rest.numel()
        return (L.pack(dx, nf, D, False, _f32((nf, D), dx.device), D, 0, rows=idx_first),
                L.pack(dx, nr, D, False, _f32((nr, D), dx.device), D, 0, rows=idx_rest), None, None)


class _SplitFrames(torch.autograd.Function):
    """tokens[:, :1], tokens[:, 1:] on 2-D rows (cvivit.py:505)"""

    @staticmethod
    def forward(ctx, x, idx_first, idx_rest):
        D = x.shape[1]
        nf, nr = idx_first.numel(), idx_rest.numel()
        ctx.save_for_backward(idx_first, idx_rest)
        ctx.M = x.shape[0]
        return (L.pack(x, nf, D, False, _f32((nf, D), x.device), D, 0, rows=idx_first),
                L.pack(x, nr, D, False, _f32((nr, D), x.device), D, 0, rows=idx_rest))

    @staticmethod
    def backward(ctx, dfirst, drest):
        idx_first, idx_rest = ctx.saved_tensors
        D = dfirst.shape[1]
        dx = _f32((ctx.M, D), dfirst.device)                        # the two row sets cover every row exactly once
        L.scatter_rows(dfirst.contiguous(), idx_first, dx, idx_first.numel(), D)
        L.scatter_rows(drest.contiguous(), idx_rest, dx, idx_rest.numel(), D)
        return dx, None, None


class _PatchEmbedFn(torch.autograd.Function):
    """Rearrange 'b c (t pt) (h p1) (w p2) -> b t h w (c pt p1 p2)' + nn.LayerNorm(P) + nn.Linear(P, dim) + nn.LayerNorm(dim) of frames
    [f0, f0 + nt pt) (cvivit.py:273-285).  The video is data: no gradient flows into it, so the first LayerNorm needs only its weight / bias
    gradients, taken against x^ recomputed from the video in the backward pass (one more 30 us read instead of a stored (rows, P) matrix)."""

    @staticmethod
    def forward(ctx, video, g1, b1, W, bW, g2, b2, geom, dtype, eps1, eps2):
        f0, nt, pt, ph, pw = geom
        B, C, F, H, Wd = video.shape
        rows = B * nt * (H // ph) * (Wd // pw)
        P = C * pt * ph * pw
        dev = video.device
        z = _f32((rows, P), dev)
        L.patchify_ln(video, f0, nt, pt, ph, pw, g1.detach(), b1.detach(), z, eps=eps1)
        y = linear_fwd(dtype, z, W.detach(), bias=bW.detach())
        out = _f32(tuple(y.shape), dev)
        L.layernorm(y, g2.detach(), b2.detach(), rows, y.shape[1], out2=out, eps=eps2)
        ctx.save_for_backward(video, W, g2, z, y)
        ctx.geom, ctx.dtype, ctx.eps1, ctx.eps2 = geom, dtype, eps1, eps2
        return out

    @staticmethod
    def backward(ctx, dout):
        video, W, g2, z, y = ctx.saved_tensors
        f0, nt, pt, ph, pw = ctx.geom
        rows, P = z.shape
        D = y.shape[1]
        dev = z.device
        dy = _f32((rows, D), dev)
        dg2, db2 = L.layernorm_bwd(y, g2.detach(), dout.contiguous(), dy, rows, D, want_beta=True, eps=ctx.eps2)
        dz, dW = linear_bwd(ctx.dtype, z, W.detach(), dy)
        dbW = L.colsum(dy, rows, D, _f32((D,), dev))
        db1 = L.colsum(dz, rows, P, _f32((P,), dev))
        xhat = _f32((rows, P), dev)
        ones, zeros = torch.ones((P,), device=dev), torch.zeros((P,), device=dev)
        L.patchify_ln(video, f0, nt, pt, ph, pw, ones, zeros, xhat, eps=ctx.eps1)
        L.mul(dz, xhat, xhat)
        dg1 = L.colsum(xhat, rows, P, _f32((P,), dev))
        return None, dg1, db1, dW, dbW, dg2, db2, None, None, None, None


class _LFQFn(torch.autograd.Function):
    """project_out(sign(project_in(x))) with the straight-through gradient of the published LFQ (`x + (quantized - x).detach()` in
    training mode; vector_quantize_pytorch, call site cvivit.py:570): d project_in(x) = d quantized.  The projection that decides the
    sign is the exact-f32 kernel of the inference path (pk_lfq_encode), so the codes of a training step are the ids the tokenizer emits."""

    @staticmethod
    def forward(ctx, x, Wp, bp, Wo, bo):
        M, D = x.shape
        cd = Wp.shape[0]
        dev = x.device
        ids = torch.empty((M,), device=dev, dtype=torch.int64)
        proj = _f32((M, cd), dev)
        L.lfq_encode(x, Wp.detach(), bp.detach(), ids, proj, M, D, cd)
        q = L.sign(proj, _f32((M, cd), dev))
        y = _f32((M, D), dev)
        L.lfq_decode(ids, Wo.detach(), bo.detach(), y, M, D, cd)
        ctx.save_for_backward(x, Wp, Wo, q)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, Wp, Wo, q = ctx.saved_tensors
        M, D = x.shape
        cd = Wp.shape[0]
        dev = x.device
        dy = dy.contiguous()
        dq, dWo = linear_bwd(L.F32, q, Wo.detach(), dy)                 # (M, cd), (D, cd)
        dbo = L.colsum(dy, M, D, _f32((D,), dev))
        dx, dWp = linear_bwd(L.F32, x, Wp.detach(), dq)                 # straight through the sign: d proj = d q
        dbp = L.colsum(dq, M, cd, _f32((cd,), dev))
        return dx, dWp, dbp, dWo, dbo


class _PatchMSE(torch.autograd.Function):
    """F.mse_loss(video, recon) (cvivit.py:591) with recon = un-patchify(pix): every pixel of the video is exactly one element of one patch row,
    so mean (video - recon)^2 = [sum_g sum (pix_g - patches_g(video))^2] / video.numel() -- the reconstruction is never laid out as a video."""

    @staticmethod
    def forward(ctx, pix_first, pix_rest, video, geoms, fmask, count):
        """fmask (B, F) uint8 or None: the frames the loss keeps (cvivit.py:585-589); count = the number of kept video elements (host float)"""
        dev = video.device
        pairs, total = [], None
        for pix, (f0, nt, pt, ph, pw) in zip((pix_first, pix_rest), geoms):
            if pix is None:
                continue
            raw = _f32(tuple(pix.shape), dev)
            L.patchify_ln(video, f0, nt, pt, ph, pw, None, None, raw)
            a = pix.detach()
            if fmask is not None:                                    # dropped frames: zero on both sides -> no loss, no gradient
                a = L.patch_frame_mask(a, _f32(tuple(pix.shape), dev), fmask, video.shape, f0, nt, pt, ph, pw)
                L.patch_frame_mask(raw, raw, fmask, video.shape, f0, nt, pt, ph, pw)
            part = L.sqdiff_sum(a.view(1, 1, 1, *a.shape), raw.view(1, 1, 1, *raw.shape))
            total = part if total is None else total + part
            pairs += [a, raw]
        ctx.save_for_backward(*pairs)
        ctx.count, ctx.has_rest = count, pix_rest is not None
        return (total / count).float()

    @staticmethod
    def backward(ctx, grad_out):
        pairs = ctx.saved_tensors
        g = grad_out.detach().float().reshape(1).contiguous()
        outs = []
        for a, raw in zip(pairs[0::2], pairs[1::2]):
            outs.append(L.scaled_diff(a, raw, 2.0 / ctx.count, _f32(tuple(a.shape), a.device), scale_dev=g))
        return outs[0], (outs[1] if ctx.has_rest else None), None, None, None, None


def _patch_embed_train(seq, video, geom, dtype):
    _, ln1, lin, ln2 = seq
    return _PatchEmbedFn.apply(video, ln1.weight, ln1.bias, lin.weight, lin.bias, ln2.weight, ln2.bias, geom, dtype, ln1.eps, ln2.eps)


def cvivit_loss_train(cv, video, *, mask=None, return_recons=False):
    """CViViT.forward (cvivit.py:518-627, use_vgg_and_gan = False) with an autograd graph over the C-ViViT parameters"""
    if cv.use_vgg_and_gan:
        raise NotImplementedError('the discriminator / VGG / adaptive-weight losses (cvivit.py:604-671) are outside the MI355X build; '
                                  'construct CViViT(use_vgg_and_gan=False) to train on the reconstruction loss')
    assert cv.lookup_free_quantization, 'the training step is built for the LFQ tokenizer (the reference default)'
    is_image = video.ndim == 4
    if is_image:
        video = video.unsqueeze(2)
    L.require_device(video, 'video')
    video = video.detach().float().contiguous()
    b, c, f, H, W = video.shape
    ph, pw = cv.patch_size
    pt = cv.temporal_patch_size
    h, w = H // ph, W // pw
    hw = h * w
    T = 1 + (f - 1) // pt
    dt = compute_dtype_of(cv)
    dev = video.device
    to_temporal, to_spatial, idx_first, idx_rest = _frame_indices(b, T, hw, dev)
    geom_first, geom_rest = (0, 1, 1, ph, pw), (1, T - 1, pt, ph, pw)

    x = _patch_embed_train(cv.to_patch_emb_first_frame, video, geom_first, dt)
    if T > 1:
        rest = _patch_embed_train(cv.to_patch_emb, video, geom_rest, dt)
        x = _MergeFrames.apply(x, rest, idx_first, idx_rest)                          # rows '(b t) (h w)'
    bias = position_bias_train(cv.spatial_rel_pos_bias, (h, w), dev)                  # one node: the encoder and the decoder both add to its gradient
    # encode: spatial, then temporal (cvivit.py:449-474)
    x = transformer_train(cv.enc_spatial_transformer, x, b * T, hw, dt, attn_bias=bias)
    x = _GatherRows.apply(x, to_temporal, to_spatial)                                 # rows '(b h w) t'
    x = transformer_train(cv.enc_temporal_transformer, x, b * hw, T, dt, video_shape=(b, T, h, w))
    # quantize (row-wise: the token order does not matter), then decode: temporal, then spatial (cvivit.py:476-516)
    vq = cv.vq
    x = _LFQFn.apply(x, vq.project_in.weight, vq.project_in.bias, vq.project_out.weight, vq.project_out.bias)
    x = transformer_train(cv.dec_temporal_transformer, x, b * hw, T, dt, video_shape=(b, T, h, w))
    x = _GatherRows.apply(x, to_spatial, to_temporal)                                 # rows '(b t) (h w)'
    x = transformer_train(cv.dec_spatial_transformer, x, b * T, hw, dt, attn_bias=bias)
    lin_first, lin_rest = cv.to_pixels_first_frame[0], cv.to_pixels[0]
    if T > 1:
        x_first, x_rest = _SplitFrames.apply(x, idx_first, idx_rest)
        pix_rest = _Linear.apply(x_rest, lin_rest.weight, lin_rest.bias, dt)
    else:
        x_first, pix_rest = x, None
    pix_first = _Linear.apply(x_first, lin_first.weight, lin_first.bias, dt)
    fmask, count = None, float(video.numel())
    if mask is not None:                                                              # variable-length training: the loss over the kept frames
        L.require_device(mask, 'mask')
        fmask = mask.to(torch.uint8).contiguous()
        count = float(mask.sum().item()) * c * H * W                                  # (one host read per step: the divisor of the mean)
    loss = _PatchMSE.apply(pix_first, pix_rest, video, (geom_first, geom_rest), fmask, count)
    if not return_recons:
        return loss
    recon = torch.empty_like(video)
    L.unpatchify(pix_first.detach(), recon, *geom_first)
    if pix_rest is not None:
        L.unpatchify(pix_rest.detach(), recon, *geom_rest)
    return loss, (recon.squeeze(2) if is_image else recon)
