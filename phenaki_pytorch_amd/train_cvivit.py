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

With `use_vgg_and_gan = True` (cvivit.py:593-671) the same graph continues into the adversarial branch (discriminator.py):

    _Unpatchify     pixel rows of both frame groups -> the reconstructed video             pk_unpatchify / pk_patchify_ln (raw rows) of the gradient
    PickFrame       one random frame per sample (pick_video_frame, cvivit.py:217-224)      pk_pick_frames, both directions
    perceptual loss F.mse_loss(vgg(frame), vgg(recon frame)) through the CALLER's `vgg=` module (cvivit.py:346-352: torchvision's pretrained VGG16
                    is the reference default and is not available offline -- any nn.Module mapping (B, 3, H, W) to features works)
    generator loss  -discr(recon frame).mean()  (hinge) through the Discriminator's HIP graph
    adaptive weight ||d perceptual / d to_pixels.weight|| / ||d gen / d to_pixels.weight||  by torch.autograd.grad over these Functions
    cvivit_discr_loss  `return_discr_loss=True` (cvivit.py:604-622): hinge loss on detached reconstructions + the gradient penalty, whose second
                    derivatives run through the same kernels (discriminator.py)
"""
import torch
import torch.nn.functional as F

from . import _lib as L
from .attention import compute_dtype_of
from .discriminator import PickFrame, bce_discr_loss, bce_gen_loss, gradient_penalty, hinge_discr_loss, hinge_gen_loss
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
    sign is the exact-f32 kernel of the inference path (pk_lfq_encode), so the codes of a training step are the ids the tokenizer emits.
    aux_cfg (dict or None): also return the module's training-mode auxiliary loss (entropy + commitment terms on project_in(x), the third
    return of `self.vq`, cvivit.py:570 -> :666) -- value and d aux / d project_in(x) come out of the pk_lfq_aux_* kernels in the forward pass."""

    @staticmethod
    def forward(ctx, x, Wp, bp, Wo, bo, aux_cfg=None, ste=True):
        """ste = False: the published module in eval() -- hard codes, no gradient reaches project_in or x (only project_out's parameters)"""
        ctx.ste = bool(ste)
        M, D = x.shape
        cd = Wp.shape[0]
        dev = x.device
        ids = torch.empty((M,), device=dev, dtype=torch.int64)
        proj = _f32((M, cd), dev)
        L.lfq_encode(x, Wp.detach(), bp.detach(), ids, proj, M, D, cd)
        q = L.sign(proj, _f32((M, cd), dev))
        y = _f32((M, D), dev)
        L.lfq_decode(ids, Wo.detach(), bo.detach(), y, M, D, cd)
        if aux_cfg is None:
            ctx.has_aux = False
            ctx.save_for_backward(x, Wp, Wo, q)
            return y
        out, dproj = L.lfq_aux(proj, **aux_cfg)
        ctx.has_aux = True
        ctx.save_for_backward(x, Wp, Wo, q, dproj)
        ctx.mark_non_differentiable(out)
        return y, out[0].clone(), out

    @staticmethod
    def backward(ctx, dy, daux=None, _dparts=None):
        x, Wp, Wo, q = ctx.saved_tensors[:4]
        M, D = x.shape
        cd = Wp.shape[0]
        dev = x.device
        dy = dy.contiguous()
        dq, dWo = linear_bwd(L.F32, q, Wo.detach(), dy)                 # (M, cd), (D, cd)
        dbo = L.colsum(dy, M, D, _f32((D,), dev))
        if not ctx.ste:
            return None, None, None, dWo, dbo, None, None
        if ctx.has_aux and daux is not None:
            dq = torch.addcmul(dq, ctx.saved_tensors[4], daux.reshape(1, 1).float())      # + d loss / d aux * d aux / d proj
        dx, dWp = linear_bwd(L.F32, x, Wp.detach(), dq)                 # straight through the sign: d proj = d q
        dbp = L.colsum(dq, M, cd, _f32((cd,), dev))
        return dx, dWp, dbp, dWo, dbo, None, None


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


class _Unpatchify(torch.autograd.Function):
    """Rearrange 'b t h w (c pt p1 p2) -> b c (t pt) (h p1) (w p2)' of both frame groups + torch.cat along time (cvivit.py:326-334, 514): the
    reconstructed video as one tensor; backward = the raw patch rows of the video gradient (the layout map is a bijection of the pixels)."""

    @staticmethod
    def forward(ctx, pix_first, pix_rest, geoms, shape):
        video = _f32(shape, pix_first.device)
        L.unpatchify(pix_first.detach().contiguous(), video, *geoms[0])
        if pix_rest is not None:
            L.unpatchify(pix_rest.detach().contiguous(), video, *geoms[1])
        ctx.geoms, ctx.shapes = geoms, (tuple(pix_first.shape), tuple(pix_rest.shape) if pix_rest is not None else None)
        return video

    @staticmethod
    def backward(ctx, dvideo):
        dvideo = dvideo.contiguous()
        outs = []
        for shape, (f0, nt, pt, ph, pw) in zip(ctx.shapes, ctx.geoms):
            if shape is None:
                outs.append(None)
                continue
            d = _f32(shape, dvideo.device)
            L.patchify_ln(dvideo, f0, nt, pt, ph, pw, None, None, d)
            outs.append(d)
        return outs[0], outs[1], None, None


def _pick_frame_indices(b, f, mask, device):
    """cvivit.py:593-602: one frame per sample, argmax of torch.randn(b, f) drawn on the HOST generator (as the reference does), masked frames excluded"""
    logits = torch.randn(b, f)
    if mask is not None:
        logits = logits.masked_fill(~mask.detach().bool().cpu(), -torch.finfo(logits.dtype).max)
    return logits.topk(1, dim=-1).indices.reshape(b).to(device=device, dtype=torch.int32)


def _gan_losses_of(cv):
    return (hinge_discr_loss, hinge_gen_loss) if cv.use_hinge_loss else (bce_discr_loss, bce_gen_loss)


def _three_channels(t):
    return t.expand(-1, 3, -1, -1) if t.shape[1] == 1 else t       # grayscale for the VGG (cvivit.py:640-641)


def cvivit_discr_loss(cv, video, *, mask=None, apply_grad_penalty=True, return_recons=False):
    """CViViT.forward(video, return_discr_loss=True) (cvivit.py:604-622): hinge (or BCE) loss of the discriminator on one random frame per sample of
    the video and of its DETACHED reconstruction, plus the gradient penalty on the real frames.  The tokenizer runs without a graph (the
    reference builds one and detaches).  apply_grad_penalty=False returns the discriminator loss alone (the reference leaves `loss` unbound there)."""
    assert cv.discr is not None, 'discriminator must exist to train it'
    is_image = video.ndim == 4
    with torch.no_grad():
        recon = cv._forward(video, mask, return_recons_only=True)
    if is_image:
        video, recon = video.unsqueeze(2), recon.unsqueeze(2)
    video = video.detach().float().contiguous()
    b, c, f = video.shape[:3]
    frame = _pick_frame_indices(b, f, mask, video.device)
    real_img = PickFrame.apply(video, frame).requires_grad_()
    fake_img = PickFrame.apply(recon.detach().contiguous(), frame)
    discr_loss_fn, _ = _gan_losses_of(cv)
    fake_logits = cv.discr(fake_img)
    real_logits = cv.discr(real_img, second_order=apply_grad_penalty)
    loss = discr_loss_fn(fake_logits, real_logits)
    if apply_grad_penalty:
        loss = loss + gradient_penalty(real_img, real_logits)
    if return_recons:
        return loss, (recon.squeeze(2) if is_image else recon)
    return loss


def _patch_embed_train(seq, video, geom, dtype):
    _, ln1, lin, ln2 = seq
    return _PatchEmbedFn.apply(video, ln1.weight, ln1.bias, lin.weight, lin.bias, ln2.weight, ln2.bias, geom, dtype, ln1.eps, ln2.eps)


def cvivit_loss_train(cv, video, *, mask=None, return_recons=False):
    """CViViT.forward (cvivit.py:518-627, use_vgg_and_gan = False) with an autograd graph over the C-ViViT parameters"""
    if cv.use_vgg_and_gan:
        assert torch.is_grad_enabled(), 'the GAN objective differentiates its terms for the adaptive weight (cvivit.py:657-662): call it with grad mode on'
        assert cv.vgg is not None, ('the perceptual loss needs a feature network: pass CViViT(vgg=<nn.Module>) -- the reference default, '
                                    "torchvision's pretrained VGG16 (cvivit.py:346-352), is not available offline")
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
    vq_aux = None
    # the published LFQ's predicate is its own `training` flag (the same predicate as quantize.LFQ.forward): in eval() it returns the hard
    # codes with NO straight-through gradient and aux = 0, also inside a grad-mode forward of a GAN-mode tokenizer
    vq_train = bool(getattr(vq, 'training', True))
    if cv.use_vgg_and_gan and hasattr(vq, 'aux_config') and vq_train:
        # the generator objective adds the quantizer's auxiliary loss (cvivit.py:570, :666); the reconstruction-only objective never reads it
        x, vq_aux, _ = _LFQFn.apply(x, vq.project_in.weight, vq.project_in.bias, vq.project_out.weight, vq.project_out.bias, vq.aux_config())
    else:
        x = _LFQFn.apply(x, vq.project_in.weight, vq.project_in.bias, vq.project_out.weight, vq.project_out.bias, None, vq_train)
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
    recon = None
    if cv.use_vgg_and_gan:
        # cvivit.py:593-602, 631-667: perceptual + adversarial terms on one random frame per sample
        recon = _Unpatchify.apply(pix_first, pix_rest, (geom_first, geom_rest), tuple(video.shape))
        frame = _pick_frame_indices(b, f, mask, dev)
        real_img = PickFrame.apply(video, frame)
        recon_img = PickFrame.apply(recon, frame)
        perceptual = F.mse_loss(cv.vgg(_three_channels(real_img)), cv.vgg(_three_channels(recon_img)))
        _, gen_loss_fn = _gan_losses_of(cv)
        gen_loss = gen_loss_fn(cv.discr(recon_img))
        if pix_rest is not None:
            last = lin_rest.weight                                                    # self.to_pixels[0].weight (cvivit.py:657)
            g_gen, = torch.autograd.grad(gen_loss, last, retain_graph=True)
            g_per, = torch.autograd.grad(perceptual, last, retain_graph=True)
            n_per, n_gen = g_per.detach().norm(p=2), g_gen.detach().norm(p=2)
            adaptive = (n_per / (n_gen + 1e-8)).clamp_(max=1e4)
            parts = cv.__dict__.get('_pk_loss_parts')
            if isinstance(parts, dict):              # tests: the terms of the objective, so the adaptive weight is pinned by itself (cvivit.py:657-664)
                parts.update(recon_loss=loss.detach().clone(), perceptual=perceptual.detach().clone(), gen_loss=gen_loss.detach().clone(),
                             norm_grad_perceptual=n_per.clone(), norm_grad_gen=n_gen.clone(), adaptive_weight=adaptive.detach().clone(),
                             grad_gen=g_gen.detach().clone(), grad_perceptual=g_per.detach().clone(),
                             vq_aux=None if vq_aux is None else vq_aux.detach().clone())
        else:
            # a 4-D image batch never reaches to_pixels (only to_pixels_first_frame): both gradient norms are 0 -> safe_div gives 0
            adaptive = torch.zeros((), device=dev)
        # cvivit.py:666: loss = recon_loss + perceptual_loss + vq_aux_loss + adaptive_weight * gen_loss (vq_aux_loss: the LFQ's entropy +
        # commitment terms, pk_lfq_aux_*; 0 for the cosine-sim VectorQuantize at inference semantics)
        loss = loss + perceptual + adaptive * gen_loss
        if vq_aux is not None:
            loss = loss + vq_aux
    if not return_recons:
        return loss
    if recon is None:
        recon = torch.empty_like(video)
        L.unpatchify(pix_first.detach(), recon, *geom_first)
        if pix_rest is not None:
            L.unpatchify(pix_rest.detach(), recon, *geom_rest)
    recon = recon.detach()
    return loss, (recon.squeeze(2) if is_image else recon.clone())
