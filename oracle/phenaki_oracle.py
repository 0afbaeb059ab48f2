"""CPU oracle: a functional, plain-torch fp32 restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- the product never imports this.

Every function takes a flat ``state_dict`` (the reference's own key names) plus a key
prefix and re-derives the arithmetic of the cited reference lines.  All paths are
/root/reference/phenaki_pytorch/<file>:<line>.  Parity is pinned by
tests/test_oracle_golden.py against tests/golden/*.pt, which oracle/make_golden.py
produced by running the real reference.
"""
import math
import torch
import torch.nn.functional as F

NEG_MAX = -torch.finfo(torch.float32).max

# --------------------------------------------------------------------------- precision mode
# 'fp32' (default): the reference's own arithmetic.
# 'bf16': the SAME algorithm with every MFMA operand of the product's bf16 mode rounded to bf16 (round-to-nearest-even)
# at exactly the points where the HIP kernels round (SURVEY.md 7 "compare against an oracle run at the same precision
# with the same rounding points"): GEMM inputs (LayerNorm outputs, the un-normalised x feeding self-attention K/V, the
# GEGLU output, the attention output, the CFG-mixed embeddings, the LayerNorm'ed patches) and weights; the attention
# operand images q^ = l2norm(q)*q_scale*scale, k^ = l2norm(k)*k_scale, v; and the softmax numerators p = exp(s - m)
# of the flash loop, tile by tile (64 keys for n >= 64 self/cross attention with >= 64 keys, else 32) with the running
# maximum the kernels use.  Accumulation, residual stream, LayerNorm statistics, softmax, position bias, PEG, LFQ and
# the critic head stay f32, as in the product.  No reference run pins this mode (the reference under bf16 autocast
# rounds at other points); it is derived from the f32 oracle, which IS pinned, by inserting roundings only.
_PRECISION = ['fp32']


class precision:
    def __init__(self, mode):
        # 'bf16x3' (the product's split-bf16 mode: three bf16 MFMAs per product, ~1e-5 per product) has no rounding points of its
        # own to mirror: it is held to the f32 arithmetic of the reference at the f32 tolerances
        assert mode in ('fp32', 'bf16', 'bf16x3')
        self.mode = 'fp32' if mode == 'bf16x3' else mode

    def __enter__(self):
        _PRECISION.append(self.mode)
        return self

    def __exit__(self, *exc):
        _PRECISION.pop()


def is_bf16():
    return _PRECISION[-1] == 'bf16'


def _r(x):
    """round to bf16 (RNE, like v_cvt_pk_bf16_f32 / torch) in bf16 mode; identity in fp32 mode"""
    return x.to(torch.bfloat16).float() if is_bf16() else x


def _lin(x, w):
    """x @ w^T with both MFMA operands in the compute type, f32 accumulation"""
    return _r(x) @ _r(w).t()


# bf16 mode: the product folds the LayerNorm in front of to_q and of the first feed-forward Linear INTO that GEMM
# (csrc/gemm.hip, pk_gemm_ex ln_s / ln_t): the MFMA operands are the rounded UN-normalised rows r(x) and the rounded folded
# weight r(gamma (.) W), the statistics are those of r(x) (taken from the operand tiles), and
#     LN(x) W^T = rstd * (r(x) r(gamma.W)^T - mean * s) + t,   s = rowsum(r(gamma.W)),  t = W beta   (f32).
# LN_FOLD mirrors phenaki_pytorch_amd.attention._LN_FOLD (PK_LN_FOLD=0 keeps the separate LayerNorm: rounding point r(LN(x))).
LN_FOLD = True
LN_FOLD_FF = False          # the feed-forward LayerNorm is folded into FF1 ...
LN_FOLD_FF_MAX_ROWS = 0     # ... for inputs of at most this many rows (0: no limit); the tests copy both from the product's settings
ATTN_FIXED_OFFSET = True    # long self-attention (n > 64, no masks): p = 2^(s log2e - integer) instead of the running-max flash loop ...
ATTN_FIXED_OFFSET_BIAS = True   # ... also when it carries a position bias (the product needs its relative-position table for that)
PATCH_FUSED = True          # bf16 patch embedding: LayerNorm(P) folded into the Linear with a centred operand (csrc/patch_embed.hip)
_ROWS_SCALE = [1]           # the product runs the cond | null halves of a CFG step as ONE batch: its row count is twice this oracle's


class _cfg_batch:
    """inside: every row count the fold rule sees is doubled (the product's 2B-sequence CFG batch)"""
    def __enter__(self):
        _ROWS_SCALE.append(2)

    def __exit__(self, *a):
        _ROWS_SCALE.pop()


def _ln_lin(x, gamma, beta, w, eps=1e-5, ff=False):
    """LayerNorm(x; gamma, beta) @ w^T -- in bf16 mode with the product's rounding points (see above)"""
    rows = _ROWS_SCALE[-1] * (x.numel() // x.shape[-1])
    ff_folded = LN_FOLD_FF and (LN_FOLD_FF_MAX_ROWS <= 0 or rows <= LN_FOLD_FF_MAX_ROWS)
    if not (is_bf16() and LN_FOLD and (ff_folded or not ff)):
        return _lin(F.layer_norm(x, x.shape[-1:], gamma, beta, eps), w)
    xb = _r(x)
    mean = xb.mean(dim=-1, keepdim=True)
    var = (xb * xb).mean(dim=-1, keepdim=True) - mean * mean
    rstd = 1.0 / torch.sqrt(var.clamp(min=0) + eps)
    wg = _r(w * gamma[None, :])
    s = wg.sum(dim=-1)
    t = w @ beta if beta is not None else torch.zeros(w.shape[0])
    return rstd * (xb @ wg.t() - mean * s) + t



# --------------------------------------------------------------------------- blocks

def gamma_layernorm(sd, p, x):
    """attention.py:29-36  LayerNorm with learned gamma and a zero ``beta`` buffer."""
    return F.layer_norm(x, x.shape[-1:], sd[p + 'gamma'], sd[p + 'beta'])


def feedforward(sd, p, x):
    """attention.py:40-53  nn.LayerNorm -> Linear(d, 2*inner, no bias) -> x*gelu(gate) -> Linear(inner, d)."""
    h = _ln_lin(x, sd[p + '0.weight'], sd[p + '0.bias'], sd[p + '1.weight'], ff=True)
    val, gate = h.chunk(2, dim=-1)
    h = F.gelu(gate) * val
    return _lin(h, sd[p + '4.weight'])


def peg(sd, p, x, shape, causal):
    """attention.py:57-85.  ``x`` (N, n, d) is reinterpreted -- raw memory order -- as
    (*shape, d) (line 73); depthwise 3x3x3 conv with zero padding, time pad (2,0) if causal
    else (1,1); result reshaped back to x's shape (line 85)."""
    orig = x.shape
    v = x.reshape(*shape, -1).permute(0, 4, 1, 2, 3)
    tpad = (2, 0) if causal else (1, 1)
    v = F.pad(v, (1, 1, 1, 1, *tpad), value=0.)
    v = F.conv3d(v, sd[p + 'dsconv.weight'], sd[p + 'dsconv.bias'], groups=v.shape[1])
    return v.permute(0, 2, 3, 4, 1).reshape(orig)


def alibi_slopes(heads):
    """attention.py:205-216"""
    def pow2(n):
        start = 2 ** (-2 ** -(math.log2(n) - 3))
        return [start * start ** i for i in range(n)]
    if math.log2(heads).is_integer():
        return pow2(heads)
    c = 2 ** math.floor(math.log2(heads))
    return pow2(c) + pow2(2 * c)[0::2][:heads - c]


def alibi_bias(heads, i, j):
    """attention.py:198-227: bias[h, a, b] = -|b - (j - i + a)| * slope_h."""
    ia = torch.arange(j - i, j)
    ja = torch.arange(j)
    bias = -(ja[None, None, :] - ia[None, :, None]).abs()
    return bias * torch.tensor(alibi_slopes(heads))[:, None, None]


def attention(sd, p, x, *, heads, causal=False, mask=None, context=None, attn_bias=None, scale=8):
    """attention.py:128-182, including its quirks: K/V of self-attention come from the
    UN-normalised x (lines 140-144); null-kv rows are interleaved k,v,k,v (line 148);
    l2norm of k happens after the null-k concat (line 153)."""
    b = x.shape[0]
    if context is not None:
        context = gamma_layernorm(sd, p + 'context_norm.', context)
    kv_in = context if context is not None else x
    q = _ln_lin(x, sd[p + 'norm.gamma'], sd[p + 'norm.beta'], sd[p + 'to_q.weight'])
    k, v = _lin(kv_in, sd[p + 'to_kv.weight']).chunk(2, dim=-1)

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], heads, -1).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)

    null_kv = sd[p + 'null_kv']                       # (h, 2*nn, dh)
    nnull = null_kv.shape[1] // 2
    nk = null_kv[:, 0::2].unsqueeze(0).expand(b, -1, -1, -1)
    nv = null_kv[:, 1::2].unsqueeze(0).expand(b, -1, -1, -1)
    k = torch.cat((nk, k), dim=-2)
    v = torch.cat((nv, v), dim=-2)

    # bf16 mode: the kernels fold the similarity scale into q^ before rounding it (a power of two commutes with the
    # rounding); q^, k^ and v are the bf16 operand images the attention kernel reads
    q = _r(F.normalize(q, dim=-1) * sd[p + 'q_scale'])
    k = _r(F.normalize(k, dim=-1) * sd[p + 'k_scale'])
    v = _r(v)
    sim = torch.einsum('bhid,bhjd->bhij', q, k) * scale
    i, j = sim.shape[-2:]
    if attn_bias is not None:
        sim = sim + F.pad(attn_bias, (nnull, 0), value=0.)
    if mask is not None:
        m = F.pad(mask, (nnull, 0), value=True)
        sim = sim.masked_fill(~m[:, None, None, :], NEG_MAX)
    if causal:
        sim = sim + alibi_bias(heads, i, j)
        cm = torch.ones((i, j), dtype=torch.bool).triu(j - i + 1)
        sim = sim.masked_fill(cm, NEG_MAX)
    fixed = ATTN_FIXED_OFFSET and (attn_bias is None or ATTN_FIXED_OFFSET_BIAS)
    if is_bf16() and fixed and context is None and nnull == 0 and mask is None and not causal and i > 64:
        # the product's fixed-offset softmax (pk_attn_fwd score_bound): an INTEGER exponent shift, so the mantissa of every p -- and
        # its bf16 rounding -- does not depend on which integer is used; numerator and row sum both use the ROUNDED p (the row sum is
        # one more MFMA block against a V^T block of ones)
        s2 = sim * 1.4426950408889634
        pt = _r(torch.exp2(s2 - torch.ceil(s2.max())))
        out = torch.einsum('bhij,bhjd->bhid', pt, v) / pt.sum(dim=-1, keepdim=True)
    elif is_bf16():
        out = _flash_bf16(sim, v, 64 if (i >= 64 and j >= 64) else 32)
    else:
        attn = sim.softmax(dim=-1)
        out = torch.einsum('bhij,bhjd->bhid', attn, v)
    out = out.permute(0, 2, 1, 3).reshape(b, i, -1)
    return _lin(out, sd[p + 'to_out.weight'])


def _flash_bf16(sim, v, tk):
    """the product's flash loop (csrc/attn.hip) restated: keys in tiles of `tk`; per tile mn = max(m, tile max),
    p = exp(s - mn) in f32; the row sum l accumulates the UN-rounded p while the PV product takes bf16(p) (f32
    accumulation); o and l are rescaled by exp(m - mn); output o / l."""
    b, h, i, j = sim.shape
    m = torch.full((b, h, i), float('-inf'))
    l = torch.zeros((b, h, i))
    o = torch.zeros((b, h, i, v.shape[-1]))
    for t0 in range(0, j, tk):
        s = sim[..., t0:t0 + tk]
        mn = torch.maximum(m, s.amax(dim=-1))
        alpha = torch.exp(m - mn)
        pt = torch.exp(s - mn[..., None])
        l = l * alpha + pt.sum(dim=-1)
        o = o * alpha[..., None] + torch.einsum('bhij,bhjd->bhid', _r(pt), v[..., t0:t0 + tk, :])
        m = mn
    return o / l[..., None]


def continuous_position_bias(sd, p, dims, nlayers=2):
    """attention.py:229-275 -> (heads, n, n)."""
    pos = [torch.arange(d) for d in dims]
    grid = torch.stack(torch.meshgrid(*pos, indexing='ij')).reshape(len(dims), -1).t()
    rel = grid[:, None, :] - grid[None, :, :]
    rel = torch.sign(rel) * torch.log(rel.abs() + 1)
    h = rel.float()
    for l in range(nlayers):
        h = F.leaky_relu(h @ sd[f'{p}net.{l}.0.weight'].t() + sd[f'{p}net.{l}.0.bias'], 0.1)
    h = h @ sd[f'{p}net.{nlayers}.weight'].t() + sd[f'{p}net.{nlayers}.bias']
    return h.permute(2, 0, 1)


def transformer(sd, p, x, *, depth, heads, causal=False, peg_on=False, peg_causal=False,
                cross=False, video_shape=None, attn_bias=None, context=None,
                self_attn_mask=None, cross_attn_context_mask=None):
    """attention.py:311-332"""
    for l in range(depth):
        lp = f'{p}layers.{l}.'
        if peg_on:
            x = peg(sd, lp + '0.', x, video_shape, peg_causal) + x
        x = attention(sd, lp + '1.', x, heads=heads, causal=causal, attn_bias=attn_bias, mask=self_attn_mask) + x
        if cross and context is not None:
            x = attention(sd, lp + '2.', x, heads=heads, context=context, mask=cross_attn_context_mask) + x
        x = feedforward(sd, lp + '3.', x) + x
    return gamma_layernorm(sd, p + 'norm_out.', x)


# --------------------------------------------------------------------------- C-ViViT

def cvivit_patch_embed(sd, cfg, video):
    """cvivit.py:273-285, 542-549: (B,C,F,H,W) -> (B,T',H',W',dim); feature order (c pt p1 p2)."""
    ph, pw = cfg['patch_size']
    pt = cfg['temporal_patch_size']
    b, c, f, H, W = video.shape
    h, w = H // ph, W // pw

    def emb(frames, tp, p):
        t = frames.shape[2] // tp
        pat = frames.reshape(b, c, t, tp, h, ph, w, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b, t, h, w, -1)
        if is_bf16() and PATCH_FUSED:
            # pk_patch_embed: operand = bf16(x - c), c = the mean of the patch's first 32 features; f32 statistics of the UNROUNDED x - c;
            # LN(x) W^T + b = rstd * (r(x - c) r(gamma.W)^T - mean' * s) + t,  s = rowsum(r(gamma.W)),  t = W beta + b
            wl, gamma, beta = sd[p + '2.weight'], sd[p + '1.weight'], sd[p + '1.bias']
            xc = pat - pat[..., :32].mean(dim=-1, keepdim=True)
            mean = xc.mean(dim=-1, keepdim=True)
            var = (xc * xc).mean(dim=-1, keepdim=True) - mean * mean
            rstd = 1.0 / torch.sqrt(var.clamp(min=0) + 1e-5)
            wg = _r(wl * gamma[None, :])
            pat = rstd * (_r(xc) @ wg.t() - mean * wg.sum(dim=-1)) + (wl @ beta + sd[p + '2.bias'])
        else:
            pat = F.layer_norm(pat, pat.shape[-1:], sd[p + '1.weight'], sd[p + '1.bias'])
            pat = _lin(pat, sd[p + '2.weight']) + sd[p + '2.bias']
        return F.layer_norm(pat, pat.shape[-1:], sd[p + '3.weight'], sd[p + '3.bias'])

    first = emb(video[:, :, :1], 1, 'to_patch_emb_first_frame.')
    if f == 1:
        return first
    rest = emb(video[:, :, 1:], pt, 'to_patch_emb.')
    return torch.cat((first, rest), dim=1)


def _spatial(sd, cfg, p, tokens):
    b, t, h, w, d = tokens.shape
    bias = continuous_position_bias(sd, 'spatial_rel_pos_bias.', (h, w))
    x = tokens.reshape(b * t, h * w, d)
    x = transformer(sd, p, x, depth=cfg['spatial_depth'], heads=cfg['heads'], attn_bias=bias,
                    video_shape=(b, t, h, w))
    return x.reshape(b, t, h, w, d)


def _temporal(sd, cfg, p, tokens):
    b, t, h, w, d = tokens.shape
    x = tokens.permute(0, 2, 3, 1, 4).reshape(b * h * w, t, d)
    # NOTE the reference passes video_shape=(b,t,h,w) although x is ((b h w), t, d)
    # (cvivit.py:456,468-470) -> the PEG sees a scrambled view; peg() reproduces it by
    # reinterpreting the contiguous buffer.
    x = transformer(sd, p, x.contiguous(), depth=cfg['temporal_depth'], heads=cfg['heads'], causal=True,
                    peg_on=True, peg_causal=True, video_shape=(b, t, h, w))
    return x.reshape(b, h, w, t, d).permute(0, 3, 1, 2, 4)


def cvivit_encode(sd, cfg, tokens):
    """cvivit.py:449-474 spatial then temporal."""
    tokens = _spatial(sd, cfg, 'enc_spatial_transformer.', tokens)
    return _temporal(sd, cfg, 'enc_temporal_transformer.', tokens)


def lfq_project(sd, x):
    """pre-sign projection of the LFQ (oracle/lfq.py); exposed for the margin audit."""
    return x @ sd['vq.project_in.weight'].t() + sd['vq.project_in.bias']


def lfq_ids(proj):
    cd = proj.shape[-1]
    mask = 2 ** torch.arange(cd - 1, -1, -1)
    return ((proj > 0).long() * mask).sum(-1)


def lfq_codes(sd, ids):
    cd = sd['vq.project_in.weight'].shape[0]
    mask = 2 ** torch.arange(cd - 1, -1, -1)
    bits = ((ids[..., None] & mask) != 0).float()
    return (bits * 2 - 1) @ sd['vq.project_out.weight'].t() + sd['vq.project_out.bias']


def cvivit_tokenize(sd, cfg, video, return_proj=False):
    """cvivit.py:518-574 with return_only_codebook_ids=True -> ids (B,T',H',W') int64."""
    tokens = cvivit_patch_embed(sd, cfg, video)
    tokens = cvivit_encode(sd, cfg, tokens)
    b, t, h, w, d = tokens.shape
    proj = lfq_project(sd, tokens.reshape(b, t * h * w, d))
    ids = lfq_ids(proj).reshape(b, t, h, w)
    return (ids, proj) if return_proj else ids


def cvivit_decode(sd, cfg, tokens):
    """cvivit.py:476-516 temporal then spatial, then to_pixels (un-patchify (c pt p1 p2))."""
    ph, pw = cfg['patch_size']
    pt = cfg['temporal_patch_size']
    H, W = cfg['image_size']
    h, w = H // ph, W // pw
    if tokens.ndim == 3:
        tokens = tokens.reshape(tokens.shape[0], -1, h, w, tokens.shape[-1])
    b = tokens.shape[0]
    tokens = _temporal(sd, cfg, 'dec_temporal_transformer.', tokens)
    tokens = _spatial(sd, cfg, 'dec_spatial_transformer.', tokens)
    c = cfg.get('channels', 3)
    first = _lin(tokens[:, :1], sd['to_pixels_first_frame.0.weight']) + sd['to_pixels_first_frame.0.bias']
    first = first.reshape(b, 1, h, w, c, 1, ph, pw).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(b, c, 1, H, W)
    if tokens.shape[1] == 1:
        return first
    rest = _lin(tokens[:, 1:], sd['to_pixels.0.weight']) + sd['to_pixels.0.bias']
    t = rest.shape[1]
    rest = rest.reshape(b, t, h, w, c, pt, ph, pw).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(b, c, t * pt, H, W)
    return torch.cat((first, rest), dim=2)


def cvivit_decode_ids(sd, cfg, ids):
    """cvivit.py:437-443 (LFQ branch)."""
    return cvivit_decode(sd, cfg, lfq_codes(sd, ids))


def cvivit_recon_loss(sd, cfg, video, mask=None):
    """cvivit.py:518-627 with use_vgg_and_gan=False: video -> ids -> decode -> F.mse_loss(video, recon), over the frames
    `mask` (b, f) keeps (variable-length training, :585-589) or over everything.  4-D image input gets a frame axis."""
    import torch.nn.functional as F
    if video.ndim == 4:
        video = video.unsqueeze(2)
    ids = cvivit_tokenize(sd, cfg, video)
    recon = cvivit_decode_ids(sd, cfg, ids.flatten(1))
    if mask is None:
        return F.mse_loss(video, recon)
    el = F.mse_loss(video, recon, reduction='none')
    return el[mask[:, None, :].expand(-1, video.shape[1], -1)].mean()


def cvivit_reconstruct_train(sd, cfg, video, return_proj=False):
    """video -> reconstruction with the module in training mode (differentiable): the LFQ output is the straight-through
    `proj + (sign(proj) - proj).detach()` of oracle/lfq.py, everything else as cvivit.py:518-583.  Autograd over the tensors of `sd`.
    return_proj: also the pre-sign projection (b, n, cd) the quantizer's auxiliary loss is taken on (oracle/lfq.py lfq_aux_loss)."""
    tokens = cvivit_patch_embed(sd, cfg, video)
    tokens = cvivit_encode(sd, cfg, tokens)
    b, t, h, w, d = tokens.shape
    proj = lfq_project(sd, tokens.reshape(b, t * h * w, d))
    q = torch.where(proj > 0, torch.ones_like(proj), -torch.ones_like(proj))
    q = proj + (q - proj).detach()
    codes = q @ sd['vq.project_out.weight'].t() + sd['vq.project_out.bias']
    recon = cvivit_decode(sd, cfg, codes)
    return (recon, proj) if return_proj else recon


def _masked_mse(video, recon, mask):
    import torch.nn.functional as F
    if mask is None:
        return F.mse_loss(video, recon)
    el = F.mse_loss(video, recon, reduction='none')
    return el[mask[:, None, :].expand(-1, video.shape[1], -1)].mean()


def cvivit_recon_loss_train(sd, cfg, video, mask=None):
    """the differentiable form of cvivit_recon_loss (cvivit.py:518-627 with the module in training mode)"""
    if video.ndim == 4:
        video = video.unsqueeze(2)
    return _masked_mse(video, cvivit_reconstruct_train(sd, cfg, video), mask)


# --------------------------------------------------------------------------- MaskGit / critic

def maskgit_embed(sd, ids):
    n = ids.shape[1]
    return sd['token_emb.weight'][ids] + sd['pos_emb.weight'][:n]


def maskgit_forward(sd, cfg, ids, *, video_patch_shape, context=None, text_mask=None,
                    video_mask=None, null_cond=False, return_embeds=False):
    """phenaki_pytorch.py:163-213.  ``null_cond`` = cond_drop_prob 1.0 -> keep mask all False
    (prob_mask_like(.., 0) -> zeros, lines 188-190)."""
    b, n = ids.shape
    if text_mask is None:
        text_mask = torch.ones((b, n), dtype=torch.bool)
    bias = continuous_position_bias(sd, 'continuous_pos_bias.', tuple(video_patch_shape))
    if null_cond:
        text_mask = torch.zeros_like(text_mask)
    x = maskgit_embed(sd, ids)
    alpha = cfg.get('gradient_shrink_alpha', 0.1)                    # :199 -- the identity in value, scales the embedding gradients
    if x.requires_grad:
        x = x * alpha + x.detach() * (1 - alpha)
    x = transformer(sd, 'transformer.', x, depth=cfg['depth'], heads=cfg['heads'], peg_on=True,
                    cross=not cfg.get('unconditional', False), video_shape=(b, *video_patch_shape),
                    attn_bias=bias, context=context, self_attn_mask=video_mask,
                    cross_attn_context_mask=text_mask)
    if return_embeds:
        return x
    return _lin(x, sd['to_logits.weight']) + sd['to_logits.bias']


def maskgit_cfg(sd, cfg, ids, *, cond_scale, **kw):
    """phenaki_pytorch.py:149-161"""
    if is_bf16() and cond_scale != 1:
        # the product mixes the 512-d trunk outputs (CFG is linear in the logits and to_logits is linear), rounds the mix to
        # bf16 and runs ONE vocab-head GEMM (csrc/sampler.hip pk_cfg_mix)
        kw = {k: v for k, v in kw.items() if k != 'return_embeds'}
        with _cfg_batch():
            e = maskgit_forward(sd, cfg, ids, null_cond=False, return_embeds=True, **kw)
            en = maskgit_forward(sd, cfg, ids, null_cond=True, return_embeds=True, **kw)
        return _lin(en + (e - en) * cond_scale, sd['to_logits.weight']) + sd['to_logits.bias']
    logits = maskgit_forward(sd, cfg, ids, null_cond=False, **kw)
    if cond_scale == 1:
        return logits
    null = maskgit_forward(sd, cfg, ids, null_cond=True, **kw)
    return null + (logits - null) * cond_scale


def critic_forward(sd, cfg, ids, *, video_patch_shape, context=None, text_mask=None,
                   video_mask=None, null_cond=False):
    """phenaki_pytorch.py:265-302 (no CPB bias, no grad shrink; head Linear(dim,1)).
    cfg['self_critic'] = (maskgit state_dict, maskgit cfg): the SelfCritic of phenaki_pytorch.py:306-336 instead -- the MaskGit
    embeddings (return_embeds=True, :334) through `to_pred` = sd['to_pred.0.weight'/'bias']."""
    if cfg.get('self_critic') is not None:
        mg_sd, mg_cfg = cfg['self_critic']
        e = maskgit_forward(mg_sd, mg_cfg, ids, video_patch_shape=video_patch_shape, context=context, text_mask=text_mask,
                            video_mask=video_mask, null_cond=null_cond, return_embeds=True)
        return (e @ sd['to_pred.0.weight'].t() + sd['to_pred.0.bias']).squeeze(-1)
    b, n = ids.shape
    if text_mask is None:
        text_mask = torch.ones((b, n), dtype=torch.bool)
    if context is not None and null_cond:
        text_mask = torch.zeros_like(text_mask)
    x = maskgit_embed(sd, ids)
    x = transformer(sd, 'transformer.', x, depth=cfg['depth'], heads=cfg['heads'], peg_on=True,
                    cross=cfg.get('has_cross_attn', False), video_shape=(b, *video_patch_shape),
                    context=context, self_attn_mask=video_mask, cross_attn_context_mask=text_mask)
    return (x @ sd['to_logits.0.weight'].t() + sd['to_logits.0.bias']).squeeze(-1)


def critic_cfg(sd, cfg, ids, *, cond_scale, **kw):
    """phenaki_pytorch.py:251-263"""
    if cond_scale == 1:
        return critic_forward(sd, cfg, ids, null_cond=False, **kw)
    with _cfg_batch():
        s = critic_forward(sd, cfg, ids, null_cond=False, **kw)
        n = critic_forward(sd, cfg, ids, null_cond=True, **kw)
    return n + (s - n) * cond_scale


# --------------------------------------------------------------------------- sampling

def mask_schedule(num_tokens, steps):
    """phenaki_pytorch.py:484-486: k_s = round(n * cos(pi/2 * s/steps)).clamp(1), fp32, for s=1..steps-1."""
    ks = [None]
    for s in range(1, steps):
        time = torch.full((1,), s / steps)
        ks.append(int((num_tokens * torch.cos(time * math.pi * 0.5)).round().long().clamp(min=1).item()))
    return ks


def gumbel_argmax(logits, temperature, u):
    """phenaki_pytorch.py:78-93 with the uniform noise ``u`` injected."""
    g = -torch.log(-torch.log(u + 1e-10) + 1e-10)
    return (logits / max(temperature, 1e-10) + g).argmax(dim=-1)


def sample_step(mg, mg_cfg, cr, cr_cfg, *, step, steps, ids, mask, scores, prime_ids, patch_shape, context,
                text_mask, cond_scale, starting_temperature, noise_K, anneal, gumbel_u, critic_u, mask_id,
                return_logits=False):
    """one iteration of the loop at phenaki_pytorch.py:478-550 (teacher-forceable)."""
    b, n = ids.shape
    out = {}
    if step > 0 and scores is not None:
        k = mask_schedule(n, steps)[step]
        idx = scores.topk(k, dim=-1).indices
        mask = torch.zeros((b, n)).scatter(1, idx, 1).bool()
    ids = torch.where(mask, mask_id, ids)
    out['masked_ids'], out['mask'] = ids.clone(), mask.clone()
    inp = ids if prime_ids is None else torch.cat((prime_ids, ids), dim=-1)
    logits = maskgit_cfg(mg, mg_cfg, inp, cond_scale=cond_scale, video_patch_shape=patch_shape,
                         context=context, text_mask=text_mask)
    if prime_ids is not None:
        logits = logits[:, prime_ids.shape[1]:]
    til = steps - (step + 1)
    temperature = starting_temperature * (til / steps)
    pred = gumbel_argmax(logits, temperature, gumbel_u)
    ids = torch.where(mask, pred, ids)
    out['pred'], out['ids'] = pred, ids
    if return_logits:
        out['logits'] = logits
    new_scores = None
    if step != steps - 1:
        if cr is not None:
            cin = ids if prime_ids is None else torch.cat((prime_ids, ids), dim=-1)
            sc = critic_cfg(cr, cr_cfg, cin, cond_scale=cond_scale, video_patch_shape=patch_shape,
                            context=context, text_mask=text_mask)
            if prime_ids is not None:
                sc = sc[:, prime_ids.shape[1]:]
            mult = {'fixed': 1., 'decay': til / steps, 'increase': (step + 1) / steps}[anneal]
            new_scores = sc + noise_K * (critic_u - 0.5) * mult
        else:
            probs = logits.softmax(dim=-1)
            sc = 1 - probs.gather(2, pred[..., None]).squeeze(-1)
            new_scores = torch.where(mask, sc, -1e4)
    out['scores'] = new_scores
    return out


def sample(cv, cv_cfg, mg, mg_cfg, cr, cr_cfg, *, num_frames, batch_size, context=None, prime_frames=None,
           steps=18, cond_scale=3., starting_temperature=0.9, noise_K=1., anneal='decay',
           noise_fn=None, trace=None, trace_logits=False):
    """phenaki_pytorch.py:418-560.  ``noise_fn(kind, step, shape)`` supplies U[0,1) noise
    ('gumbel' (B,n,V) and 'critic' (B,n)); ``trace`` (a list) receives every step's record."""
    prime_ids = None
    prime_frames_n = 0
    if prime_frames is not None:
        prime_ids = cvivit_tokenize(cv, cv_cfg, prime_frames).flatten(1)
        prime_frames_n = prime_frames.shape[2]
    H, W = cv_cfg['image_size']
    ph, pw = cv_cfg['patch_size']
    per = (H // ph) * (W // pw)
    pt = cv_cfg['temporal_patch_size']
    if prime_frames is None:
        n = per + (num_frames - 1) // pt * per
        assert (num_frames - 1) % pt == 0
    else:
        assert num_frames % pt == 0
        n = num_frames // pt * per
    text_mask = None
    if context is not None:
        text_mask = (context != 0).any(dim=-1)
    tot = num_frames + prime_frames_n
    patch_shape = (1 + (tot - 1) // pt, H // ph, W // pw)
    mask_id = mg_cfg['num_tokens']
    ids = torch.full((batch_size, n), mask_id)
    mask = torch.ones((batch_size, n), dtype=torch.bool)
    scores = None
    V = mg_cfg['num_tokens']
    for step in range(steps):
        gu = noise_fn('gumbel', step, (batch_size, n, V))
        cu = noise_fn('critic', step, (batch_size, n)) if (cr is not None and step != steps - 1) else None
        rec = sample_step(mg, mg_cfg, cr, cr_cfg, step=step, steps=steps, ids=ids, mask=mask, scores=scores,
                          prime_ids=prime_ids, patch_shape=patch_shape, context=context, text_mask=text_mask,
                          cond_scale=cond_scale, starting_temperature=starting_temperature, noise_K=noise_K,
                          anneal=anneal, gumbel_u=gu, critic_u=cu, mask_id=mask_id, return_logits=trace_logits)
        rec['temperature'] = starting_temperature * ((steps - (step + 1)) / steps)
        ids, mask, scores = rec['ids'], rec['mask'], rec['scores']
        if trace is not None:
            trace.append(rec)
    full = ids if prime_ids is None else torch.cat((prime_ids, ids), dim=-1)
    video = cvivit_decode_ids(cv, cv_cfg, full)
    if prime_frames is not None:
        video = video[:, :, prime_frames_n:]
    return video, ids


def make_video(cv, cv_cfg, mg, mg_cfg, cr, cr_cfg, *, contexts, num_frames, prime_lengths, steps=18, noise_fn_for_scene=None, traces=None, **kw):
    """phenaki_pytorch.py:691-714: one text (here: one cached context of shape (1, L_i, dim_context)) per scene; scene i+1 is primed with the
    last prime_lengths[i] frames of scene i; the last scene primes nothing.  `sample` is called with its DEFAULT cond_scale (3), as the reference
    does (:708).  ``noise_fn_for_scene(i)`` returns scene i's noise_fn; ``traces`` (a list) receives one list of step records per scene."""
    num_scenes = len(contexts)
    num_frames = tuple(num_frames) if isinstance(num_frames, (tuple, list)) else (num_frames,) * num_scenes
    prime_lengths = tuple(prime_lengths) if isinstance(prime_lengths, (tuple, list)) else (prime_lengths,) * (num_scenes - 1)
    prime_lengths = (*prime_lengths, 0)
    prime, scenes = None, []
    for i, (ctx, nf, k) in enumerate(zip(contexts, num_frames, prime_lengths)):
        tr = [] if traces is not None else None
        video, _ = sample(cv, cv_cfg, mg, mg_cfg, cr, cr_cfg, num_frames=nf, batch_size=1, context=ctx, prime_frames=prime, steps=steps,
                          noise_fn=noise_fn_for_scene(i), trace=tr, **kw)
        if traces is not None:
            traces.append(tr)
        scenes.append(video)
        prime = video[:, :, -k:]            # k = 0 -> the whole scene, as the reference's slice gives; never used (last scene)
    return torch.cat(scenes, dim=2), scenes


# --------------------------------------------------------------------------- training objective (value only)

def mask_subset_with_prob(mask, prob, perm_noise):
    """phenaki_pytorch.py:43-55 with the U[0,1) draw of its torch.rand injected as ``perm_noise`` (batch, seq)."""
    b, n = mask.shape
    num_tokens = mask.sum(dim=-1)
    num_pads = n - num_tokens
    num_masked = (prob * num_tokens).round().clamp(min=1)
    idx = perm_noise.argsort(dim=-1)
    idx = idx - num_pads[:, None]
    idx = idx.masked_fill(idx < 0, n)
    return idx < num_masked[:, None]


def phenaki_forward_loss(mg, mg_cfg, cr, cr_cfg, ids, *, patch_shape, context, steps, rand_step, perm_noise, gumbel_u,
                         mask_id, critic_loss_weight=1., critic_temperature=1., video_mask=None,
                         only_train_generator=False, only_train_critic=False):
    """Phenaki.forward (phenaki_pytorch.py:562-687), forward value only, with its three random draws injected:
    rand_step = torch.randint(0, steps, (b,)) (:620), perm_noise = the torch.rand of get_mask_subset_with_prob (:626),
    gumbel_u = the uniform noise of gumbel_sample (:653).  cond_drop_prob is 0 on this path (the reference shadows
    the argument with `cond_drop_prob = 0` at :594 before the default() at :606)."""
    import torch.nn.functional as F
    b, n = ids.shape
    text_mask = torch.any(context != 0, dim=-1) if context is not None else None              # :601
    mask_token_prob = torch.cos(rand_step * math.pi * 0.5 / steps)                              # :621
    vm = video_mask if video_mask is not None else torch.ones((b, n), dtype=torch.bool)
    mask_token_mask = mask_subset_with_prob(vm, mask_token_prob, perm_noise)
    masked_input = torch.where(mask_token_mask, mask_id, ids)
    logits = maskgit_forward(mg, mg_cfg, masked_input, video_patch_shape=patch_shape, context=context,
                             text_mask=text_mask, video_mask=video_mask)
    out = dict(mask=mask_token_mask, logits=logits)
    ce = F.cross_entropy(logits[mask_token_mask], ids[mask_token_mask])                          # :640-643
    out['ce'] = ce
    if cr is None or only_train_generator:
        out['loss'] = ce
        return out
    pred = gumbel_argmax(logits, critic_temperature, gumbel_u)                                    # :653
    critic_input = torch.where(mask_token_mask, pred, ids)
    crit = critic_forward(cr, cr_cfg, critic_input, video_patch_shape=patch_shape, context=context,
                          text_mask=text_mask, video_mask=video_mask)
    labels = (ids != pred).float()
    critic_loss = F.binary_cross_entropy_with_logits(crit, labels)                                # :673-676
    out.update(pred=pred, critic_logits=crit, critic_loss=critic_loss)
    out['loss'] = critic_loss if only_train_critic else ce + critic_loss * critic_loss_weight
    return out


def vocab_ce_grads(rows, weight, bias, targets):
    """closed-form backward of the masked-token cross entropy (phenaki_pytorch.py:640-643: F.cross_entropy(logits[mask], ids[mask]), mean
    over the R masked rows) at the vocabulary head: with p = softmax(rows W^T + b), g = (p - onehot(targets)) / R,
        d rows = g W,   d W = g^T rows,   d b = colsum(g);   also returns the loss.  f64 internally, f32 results."""
    x, w = rows.double(), weight.double()
    logits = x @ w.t() + bias.double()
    lse = torch.logsumexp(logits, dim=-1)
    R = rows.shape[0]
    loss = (lse - logits.gather(1, targets[:, None]).squeeze(1)).mean()
    g = torch.exp(logits - lse[:, None])
    g[torch.arange(R), targets] -= 1.0
    g /= R
    return dict(loss=loss.float(), d_rows=(g @ w).float(), d_weight=(g.t() @ x).float(), d_bias=g.sum(dim=0).float())
