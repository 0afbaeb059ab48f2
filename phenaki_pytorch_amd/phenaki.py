"""MaskGit / TokenCritic / SelfCritic / Phenaki / make_video with the reference's surfaces
(/root/reference/phenaki_pytorch/phenaki_pytorch.py:105-560, 691-714) on the MI355X kernels.

What changes relative to the reference's op sequence (results are the same up to f32 rounding):
  * cond and null passes of classifier-free guidance run as ONE batch of 2B sequences;
  * CFG is applied to the 512-d trunk output before the (linear) vocab head, so the 65 536-wide GEMM runs once;
  * the sampler never materialises logits: gumbel-argmax / softmax confidence live in the GEMM epilogue;
  * cross-attention K/V of the (step-invariant) text context and the position bias are computed once per sample();
  * the mask schedule k_s is data independent and precomputed on the host (no per-step .item() sync).
`Phenaki.forward` (phenaki_pytorch.py:562-687) returns the VALUE of the training objective (no autograd graph): masked-token
cross entropy without logits (pk_vocab_sample statistics + pk_vocab_ce) + the token-critic BCE.
"""
import math
from functools import partial
from typing import List

import torch
from torch import nn

from . import _lib as L
from .attention import (ContinuousPositionBias, PackedModule, Transformer, compute_dtype_of, exists, default, linear_weight,
                        param_fingerprint, set_compute_dtype)
from .cvivit import CViViT
from .t5 import t5_encode_text, get_encoded_dim, DEFAULT_T5_NAME


def cast_tuple(val, length=1):
    return val if isinstance(val, tuple) else (val,) * length


def eval_decorator(fn):
    def inner(model, *args, **kwargs):
        was_training = model.training
        model.eval()
        out = fn(model, *args, **kwargs)
        model.train(was_training)
        return out
    return inner


def uniform(shape, device):
    return torch.zeros(shape, device=device).float().uniform_(0, 1)


def prob_mask_like(shape, prob, device):
    if prob == 1:
        return torch.ones(shape, device=device, dtype=torch.bool)
    elif prob == 0:
        return torch.zeros(shape, device=device, dtype=torch.bool)
    return torch.zeros(shape, device=device).float().uniform_(0, 1) < prob


def mask_schedule(num_tokens, steps):
    """k_s = round(n * cos(pi/2 * s/steps)).clamp(1) in f32, s = 1..steps-1 (phenaki_pytorch.py:484-486), on the host."""
    ks = [None]
    for s in range(1, steps):
        time = torch.full((1,), s / steps)
        ks.append(int((num_tokens * torch.cos(time * math.pi * 0.5)).round().long().clamp(min=1).item()))
    return ks


def _u8(mask):
    if mask is None or (mask.dtype == torch.uint8 and mask.is_contiguous()):
        return mask
    return mask.to(torch.uint8).contiguous()


def _cfg_masks(text_mask, nb, with_null):
    """(nb, n_ctx) bool text mask -> uint8 mask for the S = 2 nb sequences of a CFG batch: [cond rows | all-False null rows]
    (cond_drop_prob = 1, phenaki_pytorch.py:188-190); prepared ONCE per sample() call, not per step."""
    if text_mask is None:
        return None
    tm = text_mask.to(torch.uint8)
    if not with_null:
        return tm.contiguous()
    return torch.cat((tm, torch.zeros_like(tm)), dim=0).contiguous()


class _TokenTrunk(PackedModule):
    """shared plumbing of MaskGit / TokenCritic: ids -> embeddings -> Transformer -> norm_out rows."""

    def _embed(self, ids2d, replicas=1, ids_prime=None):
        """ids2d (nb, n) [after ids_prime (nb, n_prime)] -> (replicas * nb * n_tot, D) f32: token + position embedding; the
        replicas (cond | null halves of a CFG batch) read the same id rows inside the kernel (no torch.cat)."""
        nb, n = ids2d.shape
        n_tot = n + (ids_prime.shape[-1] if ids_prime is not None else 0)
        S = replicas * nb
        D = self.token_emb.weight.shape[1]
        x = torch.empty((S * n_tot, D), device=ids2d.device, dtype=torch.float32)
        L.embed(ids2d, self.token_emb.weight, self.pos_emb.weight, x, S, n, D, nb=nb, ids_prime=ids_prime)
        return x

    def _trunk(self, ids2d, video_patch_shape, *, context=None, text_mask=None, video_mask=None, attn_bias=None,
               use_cross=True, kv_cache=None, replicas=1, ids_prime=None):
        """ids2d (nb, n) int64 -> norm_out(transformer(emb)) as (S*n_tot, D) f32 for S = replicas * nb sequences.
        context (S, n_ctx, d) f32 and the masks cover all S sequences; masks may be bool or (already prepared) uint8."""
        L.require_device(ids2d, 'token ids')
        if ids2d.dtype != torch.int64 or not ids2d.is_contiguous():
            ids2d = ids2d.long().contiguous()
        nb, n = ids2d.shape
        S = replicas * nb
        n_tot = n + (ids_prime.shape[-1] if ids_prime is not None else 0)
        ctx2, n_ctx = None, None
        if use_cross and exists(context):
            n_ctx = context.shape[1]
            ctx2 = context.reshape(S * n_ctx, context.shape[-1])
            if ctx2.dtype != torch.float32 or not ctx2.is_contiguous():
                ctx2 = ctx2.float().contiguous()
        dt = compute_dtype_of(self)
        # the cond | null copies of a CFG batch see the same ids: until the first cross-attention they are the same rows, so layer 0's
        # PEG + self-attention run on nb sequences and write both copies (Transformer.run replicas)
        shared = replicas == 2 and self.transformer.shares_cfg_prefix(dt, ctx2, video_mask)
        x = self._embed(ids2d, 1 if shared else replicas, ids_prime)
        return self.transformer.run(x, S, n_tot, dt, video_shape=(S, *video_patch_shape),
                                    attn_bias=attn_bias, context2d=ctx2, n_ctx=n_ctx, self_attn_mask=_u8(video_mask),
                                    cross_attn_context_mask=_u8(text_mask) if ctx2 is not None else None,
                                    kv_cache=kv_cache, replicas=2 if shared else 1)

    def set_compute_dtype(self, name):
        return set_compute_dtype(self, name)


class MaskGit(_TokenTrunk):
    def __init__(self, *, dim, num_tokens, max_seq_len, gradient_shrink_alpha=0.1, heads=8, dim_head=64,
                 unconditional=False, attn_dropout=0., ff_dropout=0., **kwargs):
        super().__init__()
        self.dim = dim
        self.mask_id = num_tokens
        self.unconditional = unconditional
        self.token_emb = nn.Embedding(num_tokens + 1, dim)      # last token is used as mask_id
        self.max_seq_len = max_seq_len
        self.pos_emb = nn.Embedding(max_seq_len, dim)
        self.gradient_shrink_alpha = gradient_shrink_alpha      # identity in value (forward only)
        self.continuous_pos_bias = ContinuousPositionBias(dim=dim_head, heads=heads, num_dims=3)
        self.transformer = Transformer(dim=dim, attn_num_null_kv=2, has_cross_attn=not self.unconditional,
                                       dim_head=dim_head, heads=heads, attn_dropout=attn_dropout,
                                       ff_dropout=ff_dropout, peg=True, **kwargs)
        self.to_logits = nn.Linear(dim, num_tokens)

    def _prepare(self, x, text_mask, video_patch_shape):
        assert x.ndim in {2, 4}, 'video token ids must be of shape (batch, seq) or (batch, frame, height, width)'
        if x.ndim == 4:
            video_patch_shape = x.shape[1:]
            x = x.reshape(x.shape[0], -1)
        b, n = x.shape
        assert exists(video_patch_shape), 'video patch shape must be given'
        assert n <= self.max_seq_len, f'the video token sequence length you are passing in ({n}) is greater than the `max_seq_len` ({self.max_seq_len}) set on your `MaskGit`'
        return x, tuple(video_patch_shape)

    def embeds(self, x, *, video_patch_shape, context=None, text_mask=None, video_mask=None, null_rows=0, kv_cache=None,
               replicas=1, ids_prime=None):
        """norm_out rows (S*n_tot, D) f32 for S = replicas * x.shape[0] sequences (the replicas share the id rows); context /
        masks cover all S sequences; the LAST `null_rows` sequences get an all-False text mask (cond_drop_prob = 1,
        phenaki_pytorch.py:188-190) unless the caller already prepared a uint8 mask (sample())."""
        S = replicas * x.shape[0]
        if exists(context) and not exists(text_mask):
            text_mask = torch.ones(context.shape[:2], device=x.device, dtype=torch.bool)
        if exists(text_mask) and null_rows and text_mask.dtype != torch.uint8:
            text_mask = text_mask.clone()
            text_mask[S - null_rows:] = False
        bias = self.continuous_pos_bias.spec(*video_patch_shape)       # matrix or relative-position table, as each attention launch can use
        return self._trunk(x, video_patch_shape, context=context, text_mask=text_mask, video_mask=video_mask,
                           attn_bias=bias, use_cross=not self.unconditional, kv_cache=kv_cache, replicas=replicas,
                           ids_prime=ids_prime)

    def _logits(self, e2d, rows, S, n):
        dt = compute_dtype_of(self)
        V = self.to_logits.weight.shape[0]
        out = torch.empty((rows, V), device=e2d.device, dtype=torch.float32)
        L.gemm(dt, e2d, linear_weight(self.to_logits, dt), rows, V, self.dim, C=out, bias=self.to_logits.bias)
        return out.view(S, n, V)

    @torch.no_grad()
    def forward_with_cond_scale(self, *args, cond_scale=3, **kwargs):
        if cond_scale == 1:
            return self.forward(*args, cond_drop_prob=0., **kwargs)
        x = args[0]
        kwargs = dict(kwargs)
        x, vps = self._prepare(x, kwargs.get('text_mask'), kwargs.pop('video_patch_shape', None))
        b, n = x.shape
        context, text_mask, video_mask = kwargs.get('context'), kwargs.get('text_mask'), kwargs.get('video_mask')
        rep = lambda t: None if t is None else torch.cat((t, t), dim=0)
        e = self.embeds(x, replicas=2, video_patch_shape=vps, context=rep(context), text_mask=rep(text_mask),
                        video_mask=rep(video_mask), null_rows=b)
        dt = compute_dtype_of(self)
        mixed = torch.empty((b * n, self.dim), device=x.device, dtype=L.tdtype(dt))
        L.cfg_mix(e, b, n, 0, None, b * n, float(cond_scale), True, mixed, self.dim)
        return self._logits(mixed, b * n, b, n)

    def forward(self, x, cond_drop_prob=0., text_mask=None, video_mask=None, video_patch_shape=None,
                return_embeds=False, **kwargs):
        """phenaki_pytorch.py:163-213.  Grad mode on + trainable parameters: the logits / embeddings carry an autograd graph over this
        library's backward kernels (train.py); otherwise the fused inference path."""
        from .train import maskgit_forward_train, wants_grad
        if wants_grad(self):
            context = kwargs.pop('context', None)
            assert not kwargs, f'unexpected arguments {sorted(kwargs)}'
            return maskgit_forward_train(self, x, cond_drop_prob=cond_drop_prob, text_mask=text_mask, video_mask=video_mask,
                                         video_patch_shape=video_patch_shape, return_embeds=return_embeds, context=context)
        with torch.no_grad():
            return self._forward_value(x, cond_drop_prob, text_mask, video_mask, video_patch_shape, return_embeds, **kwargs)

    def _forward_value(self, x, cond_drop_prob=0., text_mask=None, video_mask=None, video_patch_shape=None,
                       return_embeds=False, **kwargs):
        x, vps = self._prepare(x, text_mask, video_patch_shape)
        b, n = x.shape
        context = kwargs.pop('context', None)
        assert not kwargs, f'unexpected arguments {sorted(kwargs)}'
        if exists(context) and not exists(text_mask):
            text_mask = torch.ones(context.shape[:2], device=x.device, dtype=torch.bool)
        if cond_drop_prob > 0 and exists(text_mask):
            keep_mask = prob_mask_like((b,), 1 - cond_drop_prob, device=x.device)
            text_mask = keep_mask[:, None] & text_mask
        e = self.embeds(x, video_patch_shape=vps, context=context, text_mask=text_mask, video_mask=video_mask)
        if return_embeds:
            return e.view(b, n, self.dim)
        return self._logits(e, b * n, b, n)


class TokenCritic(_TokenTrunk):
    def __init__(self, *, dim, num_tokens, max_seq_len, has_cross_attn=False, attn_dropout=0., ff_dropout=0., **kwargs):
        super().__init__()
        self.has_cross_attn = has_cross_attn
        self.mask_id = num_tokens
        self.token_emb = nn.Embedding(num_tokens + 1, dim)
        self.pos_emb = nn.Embedding(max_seq_len, dim)
        self.transformer = Transformer(dim=dim, peg=True, attn_dropout=attn_dropout, ff_dropout=ff_dropout,
                                       has_cross_attn=has_cross_attn, **kwargs)
        self.to_logits = nn.Sequential(nn.Linear(dim, 1), nn.Identity())

    def head(self):
        lin = self.to_logits[0]
        return lin.weight.reshape(-1), lin.bias

    def embeds(self, x, *, video_patch_shape, context=None, text_mask=None, video_mask=None, null_rows=0, kv_cache=None,
               replicas=1, ids_prime=None):
        S = replicas * x.shape[0]
        if exists(context) and not exists(text_mask):
            text_mask = torch.ones(context.shape[:2], device=x.device, dtype=torch.bool)
        if exists(text_mask) and exists(context) and null_rows and text_mask.dtype != torch.uint8:
            text_mask = text_mask.clone()
            text_mask[S - null_rows:] = False
        return self._trunk(x, video_patch_shape, context=context, text_mask=text_mask, video_mask=video_mask,
                           use_cross=self.has_cross_attn, kv_cache=kv_cache, replicas=replicas, ids_prime=ids_prime)

    def _scores(self, x, video_patch_shape, context, text_mask, video_mask, cond_scale, with_null):
        b, n = x.shape
        rep = lambda t: None if t is None else torch.cat((t, t), dim=0)
        if with_null:
            e = self.embeds(x, replicas=2, video_patch_shape=video_patch_shape, context=rep(context),
                            text_mask=rep(text_mask), video_mask=rep(video_mask), null_rows=b)
        else:
            e = self.embeds(x, video_patch_shape=video_patch_shape, context=context, text_mask=text_mask, video_mask=video_mask)
        w, bias = self.head()
        out = torch.empty((b, n), device=x.device, dtype=torch.float32)
        L.critic_head(e, w, bias, e.shape[1], b, n, 0, with_null, float(cond_scale), None, 0., out)
        return out

    @staticmethod
    def _flatten(x, video_patch_shape):
        if exists(video_patch_shape):
            vps = tuple(video_patch_shape)
        else:
            vps = tuple(x.shape[1:])
        return x.reshape(x.shape[0], -1), vps

    @torch.no_grad()
    def forward_with_cond_scale(self, *args, cond_scale=3, **kwargs):
        kwargs = dict(kwargs)
        x, vps = self._flatten(args[0], kwargs.pop('video_patch_shape', None))
        context = kwargs.get('context')
        with_null = cond_scale != 1 and exists(context)      # without context both passes are identical
        return self._scores(x, vps, context, kwargs.get('text_mask'), kwargs.get('video_mask'), cond_scale, with_null)

    def forward(self, x, text_mask=None, cond_drop_prob=None, context=None, video_mask=None, video_patch_shape=None, **kwargs):
        """phenaki_pytorch.py:265-302; under grad mode with trainable parameters: with an autograd graph (train.py)"""
        from .train import critic_forward_train, wants_grad
        if wants_grad(self):
            return critic_forward_train(self, x, text_mask=text_mask, cond_drop_prob=cond_drop_prob, context=context, video_mask=video_mask,
                                        video_patch_shape=video_patch_shape)
        with torch.no_grad():
            return self._forward_value(x, text_mask, cond_drop_prob, context, video_mask, video_patch_shape)

    def _forward_value(self, x, text_mask=None, cond_drop_prob=None, context=None, video_mask=None, video_patch_shape=None):
        x, vps = self._flatten(x, video_patch_shape)
        b = x.shape[0]
        if exists(context) and not exists(text_mask):
            text_mask = torch.ones(context.shape[:2], device=x.device, dtype=torch.bool)
        if exists(context) and exists(cond_drop_prob) and cond_drop_prob > 0:
            keep_mask = prob_mask_like((b,), 1 - cond_drop_prob, device=x.device)
            text_mask = keep_mask[:, None] & text_mask
        return self._scores(x, vps, context, text_mask, video_mask, 1., False)


class SelfCritic(PackedModule):
    """phenaki_pytorch.py:306-336 : MaskGit embeddings -> Linear(dim, 1)."""

    def __init__(self, maskgit: MaskGit):
        super().__init__()
        self.maskgit = maskgit
        self.to_pred = nn.Sequential(nn.Linear(maskgit.dim, 1), nn.Identity())
        self.has_cross_attn = not maskgit.unconditional

    def head(self):
        lin = self.to_pred[0]
        return lin.weight.reshape(-1), lin.bias

    def embeds(self, x, **kw):
        return self.maskgit.embeds(x, **kw)

    @torch.no_grad()
    def forward_with_cond_scale(self, *args, cond_scale=3, **kwargs):
        kwargs = dict(kwargs)
        x, vps = self.maskgit._prepare(args[0], kwargs.get('text_mask'), kwargs.pop('video_patch_shape', None))
        b, n = x.shape
        context, text_mask, video_mask = kwargs.get('context'), kwargs.get('text_mask'), kwargs.get('video_mask')
        with_null = cond_scale != 1
        rep = lambda t: None if t is None else torch.cat((t, t), dim=0)
        if with_null:
            e = self.maskgit.embeds(x, replicas=2, video_patch_shape=vps, context=rep(context),
                                    text_mask=rep(text_mask), video_mask=rep(video_mask), null_rows=b)
        else:
            e = self.maskgit.embeds(x, video_patch_shape=vps, context=context, text_mask=text_mask, video_mask=video_mask)
        w, bias = self.head()
        out = torch.empty((b, n), device=x.device, dtype=torch.float32)
        L.critic_head(e, w, bias, e.shape[1], b, n, 0, with_null, float(cond_scale), None, 0., out)
        return out

    def forward(self, x, *args, **kwargs):
        """phenaki_pytorch.py:334-336; under grad mode with trainable parameters: with an autograd graph (train.py)"""
        from .train import critic_forward_train, wants_grad
        if wants_grad(self):
            names = ('cond_drop_prob', 'text_mask', 'video_mask', 'video_patch_shape')
            kw = dict(zip(names, args))
            kw.update(kwargs)
            return critic_forward_train(self, x, text_mask=kw.get('text_mask'), cond_drop_prob=kw.get('cond_drop_prob'), context=kw.get('context'),
                                        video_mask=kw.get('video_mask'), video_patch_shape=kw.get('video_patch_shape'))
        with torch.no_grad():
            return self._forward_value(x, *args, **kwargs)

    def _forward_value(self, x, *args, **kwargs):
        embeds = self.maskgit(x, *args, return_embeds=True, **kwargs)
        b, n, d = embeds.shape
        w, bias = self.head()
        out = torch.empty((b, n), device=x.device, dtype=torch.float32)
        L.critic_head(embeds.reshape(b * n, d), w, bias, d, b, n, 0, False, 1., None, 0., out)
        return out


class Phenaki(PackedModule):
    def __init__(self, *, maskgit: MaskGit, cvivit: CViViT, critic=None, steps=18, t5_name=DEFAULT_T5_NAME,
                 sample_temperature=0., text_embed_dim=None, cond_drop_prob=0.25, max_text_len=128,
                 self_token_critic=False, critic_loss_weight=1., critic_noise_anneal_schedule='decay',
                 critic_train_sample_temperature=1.):
        super().__init__()
        assert isinstance(maskgit, MaskGit) and isinstance(cvivit, CViViT)
        assert critic is None or isinstance(critic, (TokenCritic, SelfCritic))
        self.cvivit = cvivit.copy_for_eval()
        self.maskgit = maskgit
        self.unconditional = maskgit.unconditional
        self.mask_id = maskgit.mask_id
        assert not (self_token_critic and exists(critic))
        if self_token_critic:
            critic = SelfCritic(maskgit)
        if exists(critic):
            critic = critic.eval()
        assert not exists(critic) or self_token_critic or (not maskgit.unconditional) == critic.has_cross_attn
        self.critic = critic
        self.critic_noise_anneal_schedule = critic_noise_anneal_schedule
        self.critic_loss_weight = critic_loss_weight
        self.critic_train_sample_temperature = critic_train_sample_temperature
        self.steps = steps
        self.sample_temperature = sample_temperature
        text_embed_dim = default(text_embed_dim, get_encoded_dim(t5_name) if text_embed_dim is None else None)
        self.encode_texts = partial(t5_encode_text, name=t5_name)
        self.text_embed_dim = text_embed_dim
        self.max_text_len = max_text_len
        assert cond_drop_prob > 0.
        self.cond_drop_prob = cond_drop_prob

    def set_compute_dtype(self, name):
        return set_compute_dtype(self, name)

    def sample_images(self, *, texts=None, batch_size=1, cond_scale=3., starting_temperature=0.9, noise_K=1.):
        video = self.sample(texts=texts, num_frames=1, cond_scale=cond_scale, starting_temperature=starting_temperature,
                            noise_K=noise_K)
        return video.squeeze(2)

    def forward(self, *args, **kwargs):
        """phenaki_pytorch.py:562-687.  With grad mode on and trainable MaskGit / critic parameters this is the training step: the loss
        carries an autograd graph whose nodes are the MI355X forward / backward kernels (train.py::phenaki_loss), `loss.backward()` fills
        the .grad of the MaskGit and critic parameters (the C-ViViT and the text encoder are frozen here as in the reference, :580-598).
        Otherwise (no_grad / eval without trainable parameters): the value only, through the fused inference kernels."""
        if torch.is_grad_enabled() and any(p.requires_grad for m in (self.maskgit, self.critic) if exists(m) for p in m.parameters()):
            from .train import phenaki_loss
            return phenaki_loss(self, *args, **kwargs)
        return self.objective_value(*args, **kwargs)

    @torch.no_grad()
    def objective_value(self, videos=None, *, texts=None, video_codebook_ids=None, video_frame_mask=None, text_embeds=None,
                        cond_drop_prob=None, only_train_generator=False, only_train_critic=False, _draws=None):
        """phenaki_pytorch.py:562-687 -- the VALUE of the training objective (masked-token cross entropy + weighted
        token-critic BCE).  Forward only: the result carries no autograd graph (the backward kernels are SURVEY.md 8f
        "next").  The 65 536-way logits are never written: pk_vocab_sample draws the critic's input ids and leaves the
        softmax statistics, pk_vocab_ce turns them into per-row losses.  As in the reference, `cond_drop_prob` has no
        effect on this path (it is shadowed by `cond_drop_prob = 0` at :594).
        _draws (tests): dict(rand_step (b,), perm_noise (b,n) U[0,1), gumbel_u (b,n,V) U[0,1)) replaces the three
        random draws of the reference (:620, :626 -> :43-55, :653)."""
        assert not (only_train_generator and only_train_critic)
        assert not (only_train_critic and not exists(self.critic)), 'only_train_critic needs a critic (Phenaki(critic=...) or self_token_critic=True)'
        assert exists(videos) ^ exists(video_codebook_ids), 'either raw video or video codebook ids must be given'
        assert not (exists(videos) and not exists(self.cvivit)), 'cvivit must be provided if one wants to encode the videos live during training'
        assert (exists(text_embeds) ^ exists(texts)) ^ self.unconditional, \
            'either raw text of text embeds must be given, and if unconditional, none should be given'
        assert not (exists(text_embeds) and text_embeds.shape[-1] != self.text_embed_dim), 'text embedding dimension is not correct'
        mg, critic = self.maskgit, self.critic
        if not exists(video_codebook_ids):
            assert videos.ndim in {4, 5}
            if videos.ndim == 4:
                videos = videos.unsqueeze(2)
            video_codebook_ids = self.cvivit(videos, return_only_codebook_ids=True)
        L.require_device(video_codebook_ids, 'video_codebook_ids')
        assert video_codebook_ids.ndim == 4, 'video codebook ids must be (batch, frames, height, width): MaskGit takes the patch shape from it'
        device = video_codebook_ids.device
        patch_shape = tuple(video_codebook_ids.shape[1:])

        text_mask = None
        if not self.unconditional:
            if not exists(text_embeds):
                text_embeds = self.encode_texts(texts, output_device=device)
            text_embeds = text_embeds.to(device).float()
            text_mask = torch.any(text_embeds != 0, dim=-1)

        video_mask = None
        if exists(video_frame_mask):
            video_mask = self.cvivit.calculate_video_token_mask(videos, video_frame_mask=video_frame_mask)

        ids = video_codebook_ids.reshape(video_codebook_ids.shape[0], -1).long().contiguous()
        b, n = ids.shape
        draws = _draws or {}
        rand_step = draws['rand_step'].to(device) if 'rand_step' in draws else torch.randint(0, self.steps, (b,), device=device)
        mask_token_prob = torch.cos(rand_step * math.pi * 0.5 / self.steps)
        vm = video_mask if exists(video_mask) else torch.ones((b, n), device=device, dtype=torch.bool)
        # get_mask_subset_with_prob (phenaki_pytorch.py:43-55)
        num_tokens = vm.sum(dim=-1)
        num_masked = (mask_token_prob * num_tokens).round().clamp(min=1)
        perm_noise = draws['perm_noise'].to(device) if 'perm_noise' in draws else torch.rand((b, n), device=device)
        perm = perm_noise.argsort(dim=-1) - (n - num_tokens)[:, None]
        perm = perm.masked_fill(perm < 0, n)
        mask_token_mask = perm < num_masked[:, None]
        masked_input = torch.where(mask_token_mask, self.mask_id, ids)

        dt = compute_dtype_of(mg)
        D, V = mg.dim, mg.to_logits.weight.shape[0]
        e = mg.embeds(masked_input, video_patch_shape=patch_shape, context=text_embeds, text_mask=text_mask, video_mask=video_mask)
        # rows of the vocab head: the critic's labels compare the ids with the gumbel-sampled prediction at EVERY position
        # (phenaki_pytorch.py:653,671), so with a critic the head runs on all b*n rows; the generator-only objective needs
        # the masked rows only (>= 1 per sample, clamp(min=1) above)
        need_critic = exists(critic) and not only_train_generator
        flat_mask = mask_token_mask.reshape(-1)
        rows = None if need_critic else flat_mask.nonzero().reshape(-1).int()
        M = b * n if need_critic else rows.numel()
        mixed = torch.empty((M, D), device=device, dtype=L.tdtype(dt))
        L.cfg_mix(e, b, n, 0, rows, M, 1.0, False, mixed, D)                    # (gather the rows,) cast to T
        w_logits = linear_weight(mg.to_logits, dt)
        partials = torch.empty((5 * L.vocab_ntiles(V) * M,), device=device, dtype=torch.float32)
        U = draws['gumbel_u'].to(device).float().contiguous() if 'gumbel_u' in draws else None
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if U is None else 0
        L.vocab_sample(dt, mixed, w_logits, mg.to_logits.bias, M, V, D, float(self.critic_train_sample_temperature), U, rows,
                       seed, True, partials)
        loss = None
        if not only_train_critic:
            loss_rows = torch.empty((M,), device=device, dtype=torch.float32)
            L.vocab_ce(dt, partials, M, V, mixed, w_logits, mg.to_logits.bias, D, ids.reshape(-1), rows, loss_rows)
            loss = (loss_rows[flat_mask] if need_critic else loss_rows).mean()
        if not need_critic:
            return loss
        pred = torch.empty((b * n,), device=device, dtype=torch.long)
        L.vocab_reduce(partials, M, V, None, None, None, pred, None, False)
        pred = pred.view(b, n)
        critic_input = torch.where(mask_token_mask, pred, ids)
        crit = critic(critic_input.view(b, *patch_shape), video_mask=video_mask, cond_drop_prob=0., text_mask=text_mask,
                      context=text_embeds)
        labels = (ids != pred).float()
        critic_loss = torch.nn.functional.binary_cross_entropy_with_logits(crit.reshape(b, n), labels)
        if only_train_critic:
            return critic_loss
        return loss + critic_loss * self.critic_loss_weight

    # ------------------------------------------------------------------------------------------ sampling (phenaki_pytorch.py:418-560)

    def enable_sample_graph(self, on=True, static_output=False):
        """replay the whole 18-step sampling loop + final decode as ONE captured hipGraph per (batch, frames, prime, context
        length, guidance, temperature) configuration: ~3 000 kernel launches become one graph launch.  Noise stays fresh per
        call (the kernels add a device-resident seed word to their captured seeds).  Eager launches remain the default and the
        only mode for the parity hooks (_noise_fn / _trace).
        static_output=True: `sample` returns the graph's OWN output buffer (overwritten by the next call with the same configuration)
        instead of a clone of it -- saves a 13.4 MB-per-video copy per call for callers that consume the video before sampling again."""
        self.__dict__['_pk_sample_graph'] = bool(on)
        self.__dict__['_pk_sample_graph_static'] = bool(on and static_output)
        if not on:
            self.__dict__.pop('_pk_sample_graphs', None)
        return self

    def _sample_loop(self, st):
        """the mask-predict loop on prepared state `st` (device buffers + host scalars); kernels and allocations only, no host
        synchronisation and no torch elementwise ops: capturable as a hipGraph."""
        mg, critic = self.maskgit, self.critic
        dt = compute_dtype_of(mg)
        B, n, n_tot, npr, V, D = st['B'], st['n'], st['n_tot'], st['n_prime'], st['V'], st['D']
        ids, mask, pred, scores2 = st['ids'], st['mask'], st['pred'], st['scores']
        rows_buf, partials, mixed = st['rows'], st['partials'], st['mixed']
        ks, steps = st['ks'], self.steps
        noise_fn, trace, seed_dev = st['noise_fn'], st['trace'], st['seed_dev']
        seed_base = st['seed_base'] if seed_dev is None else 0          # graph mode: the base seed lives in *seed_dev
        M64 = 0xFFFFFFFFFFFFFFFF
        with_null, c_null = st['with_null'], st['c_null']
        need_lse = not exists(critic)
        mg_cache, cr_cache = {}, {}
        w_logits = linear_weight(mg.to_logits, dt)
        have_scores = False
        for step in range(steps):
            is_last_step = step == (steps - 1)
            steps_til_x0 = steps - (step + 1)
            cur, nxt = scores2[step & 1], scores2[(step + 1) & 1]

            rows, M = None, B * n
            if step > 0 and have_scores:
                # mask + ids = where(mask, mask_id, ids); only the k_s re-masked positions need the vocab head this step
                # (predictions elsewhere are discarded, phenaki_pytorch.py:509) -> compact row list for the head
                if st['compact']:
                    rows, M = rows_buf, B * ks[step]
                L.topk_mask(cur, B, n, ks[step], self.mask_id, mask, ids, rows, scores_next=nxt if need_lse else None)

            if st.get('force_fn') is not None:
                # parity hook (eager only): the caller overwrites this step's masked ids / mask in place -- teacher forcing from a
                # reference run's recorded inputs (tests: all 18 steps audited even after a near-tie changed a free-running input)
                st['force_fn'](step, ids, mask)

            rec = None
            if trace is not None:
                rec = dict(step=step, masked_ids=ids.clone(), mask=mask.bool().clone(), prime_ids=st['prime_ids'])

            e = mg.embeds(ids, replicas=2 if with_null else 1, ids_prime=st['prime_ids'], video_patch_shape=st['patch_shape'],
                          context=st['ctx_r'], text_mask=st['tm_r'], kv_cache=mg_cache)
            L.cfg_mix(e, B, n_tot, npr, rows, M, float(st['cond_scale']), with_null, mixed, D)

            temperature = st['starting_temperature'] * (steps_til_x0 / steps)
            U = noise_fn('gumbel', step, (B, n, V)) if noise_fn is not None else None
            if isinstance(U, L.TorchPhilox):
                # torch's own RNG stream: the noise torch.zeros((B, n, V)).uniform_() would hold, generated in the epilogue; then the
                # generator moves on as if that fill had happened (the reference's gumbel_noise, phenaki_pytorch.py:88-93)
                L.vocab_sample_philox(dt, mixed, w_logits, mg.to_logits.bias, M, V, D, float(temperature), rows, U, need_lse, partials)
                U.advance()
            else:
                L.vocab_sample(dt, mixed, w_logits, mg.to_logits.bias, M, V, D, float(temperature), U, rows,
                               (seed_base + step * 0x9E3779B97F4A7C15) & M64, need_lse, partials, seed_dev=seed_dev)
            L.vocab_reduce(partials, M, V, rows, mask, ids, pred, nxt if need_lse else None, need_lse)
            if rec is not None:
                rec.update(pred=pred.clone(), ids=ids.clone())

            if not is_last_step:
                if exists(critic):
                    ce = critic.embeds(ids, replicas=2 if c_null else 1, ids_prime=st['prime_ids'], video_patch_shape=st['patch_shape'],
                                       context=st['c_ctx_r'], text_mask=st['c_tm_r'], kv_cache=cr_cache)
                    if self.critic_noise_anneal_schedule == 'fixed':
                        noise_multiplier = 1.
                    elif self.critic_noise_anneal_schedule == 'decay':
                        noise_multiplier = steps_til_x0 / steps
                    elif self.critic_noise_anneal_schedule == 'increase':
                        noise_multiplier = (step + 1) / steps
                    else:
                        raise ValueError('invalid critic noise anneal schedule name')
                    u = noise_fn('critic', step, (B, n)).contiguous() if noise_fn is not None else None
                    w, bias = critic.head()
                    L.critic_head(ce, w, bias, ce.shape[1], B, n_tot, npr, c_null, float(st['cond_scale']), u,
                                  float(st['noise_K'] * noise_multiplier), nxt,
                                  seed=(seed_base + (2 * step + 1) * 0xD6E8FEB86659FD93) & M64, seed_dev=seed_dev)
                have_scores = True
                if rec is not None:
                    rec['scores'] = nxt.clone()
            if rec is not None:
                trace.append(rec)

        video = self.cvivit.decode_from_codebook_indices(ids, _prime_indices=st['prime_ids'])
        if st['prime_ids'] is not None:
            video = video[:, :, st['prime_num_frames']:]
        return video

    def _sample_state(self, B, n, n_prime, device, dt):
        V = self.maskgit.to_logits.weight.shape[0]
        D = self.maskgit.dim
        return dict(B=B, n=n, n_prime=n_prime, n_tot=n + n_prime, V=V, D=D,
                    ids=torch.empty((B, n), device=device, dtype=torch.int64),
                    mask=torch.empty((B, n), device=device, dtype=torch.uint8),
                    pred=torch.empty((B, n), device=device, dtype=torch.int64),
                    scores=[torch.empty((B, n), device=device, dtype=torch.float32) for _ in range(2)],
                    rows=torch.empty((B * n,), device=device, dtype=torch.int32),
                    partials=torch.empty((5 * L.vocab_ntiles(V) * B * n,), device=device, dtype=torch.float32),
                    mixed=torch.empty((B * n, D), device=device, dtype=L.tdtype(dt)))

    @eval_decorator
    @torch.no_grad()
    def sample(self, *, num_frames, texts=None, prime_frames=None, batch_size=1, cond_scale=3.,
               starting_temperature=0.9, noise_K=1., _noise_fn=None, _trace=None, _return_ids=False, _compact=None, _seed=None,
               _force_fn=None, _prime_ids=None):
        """phenaki_pytorch.py:418-560.  `_noise_fn(kind, step, shape)` (tests) injects the U[0,1) draws of the
        reference run ('gumbel' (B,n,V) and 'critic' (B,n), device f32 tensors); _noise_fn='torch': torch's device generator in the
        reference's order (a seeded reference run on the same GPU draws the same noise); without it the noise comes from the
        in-kernel counter hash seeded from torch's default (CPU) generator (`_seed`: the exact 64-bit stream seed instead;
        the seed a call used is kept in `self._pk_last_seed`).  `_force_fn(step, ids, mask)` (tests) may overwrite a step's masked
        input ids (B, n) int64 and mask (B, n) uint8 in place before the trunk runs (teacher forcing; eager, no row compaction);
        `_prime_ids` (tests, with prime_frames given) replaces the token ids the tokenizer would produce for the prime frames."""
        device = next(self.parameters()).device
        L.require_device(next(self.parameters()), 'Phenaki parameters')
        mg, critic = self.maskgit, self.critic
        dt = compute_dtype_of(mg)
        if isinstance(_noise_fn, str):
            # _noise_fn = 'torch': the reference's own noise -- torch's device generator, consumed in the reference's order (per step one
            # uniform_ of (B, n, V) for gumbel_sample, phenaki_pytorch.py:88-93, then one of (B, n) for the critic scores, :69-70 / :541-543).
            # The (B, n, V) fill is never materialised: the vocabulary head reproduces its elements from the Philox state (pk_vocab_sample_philox)
            # and the generator is moved past it, so `torch.manual_seed(s); phenaki.sample(...)` draws what a reference run on this GPU draws.
            assert _noise_fn == 'torch', "_noise_fn: a callable or 'torch'"

            def _noise_fn(kind, step, shape):
                if kind == 'gumbel':
                    return L.TorchPhilox(device, shape[0] * shape[1] * shape[2])
                return torch.zeros(shape, device=device).float().uniform_(0, 1)

        has_prime = exists(prime_frames)
        prime_token_ids = None
        prime_token_length = 0
        prime_num_frames = 0
        if has_prime:
            prime_token_ids = self.cvivit(prime_frames, return_only_codebook_ids=True)
            prime_token_ids = prime_token_ids.reshape(prime_token_ids.shape[0], -1).contiguous()
            if _prime_ids is not None:
                assert tuple(_prime_ids.shape) == tuple(prime_token_ids.shape)
                prime_token_ids = _prime_ids.to(device=device, dtype=prime_token_ids.dtype).contiguous()
            prime_token_length = prime_token_ids.shape[-1]
            prime_num_frames = prime_frames.shape[2]

        num_tokens = self.cvivit.num_tokens_per_frames(num_frames, include_first_frame=not has_prime)

        text_embeds = text_mask = None
        if exists(texts):
            if isinstance(texts, str):
                texts = [texts]
            text_embeds = self.encode_texts(texts, output_device=device)
            text_embeds = text_embeds.to(device).float().contiguous()
            text_mask = torch.any(text_embeds != 0, dim=-1)
            batch_size = len(texts)

        patch_shape = self.cvivit.get_video_patch_shape(num_frames + prime_num_frames, include_first_frame=True)
        B, n = batch_size, num_tokens

        # classifier-free guidance: the cond and null passes run as ONE batch of 2B sequences; context / masks of that batch
        # are laid out once per call (the null half = an all-False text mask, phenaki_pytorch.py:188-190)
        has_ctx = exists(text_embeds) and not mg.unconditional
        with_null = has_ctx and cond_scale != 1
        rep = lambda t: torch.cat((t, t), dim=0) if with_null else t
        c_has_ctx = exists(critic) and exists(text_embeds) and critic.has_cross_attn
        c_null = (c_has_ctx and cond_scale != 1) if not isinstance(critic, SelfCritic) else cond_scale != 1
        crep = lambda t: torch.cat((t, t), dim=0) if c_null else t

        seed_base = int(torch.randint(0, 2 ** 62, (1,)).item()) if _noise_fn is None else 0
        if _noise_fn is None and torch.distributed.is_available() and torch.distributed.is_initialized():
            # batch-sharded sampling: every rank draws from its own noise stream even under a common torch seed
            seed_base = (seed_base + 0xD1B54A32D192ED03 * (torch.distributed.get_rank() + 1)) & 0x3FFFFFFFFFFFFFFF
        if _seed is not None:
            seed_base = int(_seed) & 0x3FFFFFFFFFFFFFFF
        self.__dict__['_pk_last_seed'] = seed_base
        compact = (_trace is None) if _compact is None else bool(_compact)    # traces record the prediction at EVERY position
        if _force_fn is not None:
            compact = False                                                    # the forced mask need not be the top-k one

        use_graph = self.__dict__.get('_pk_sample_graph', False) and _noise_fn is None and _trace is None and _force_fn is None
        key = (B, n, prime_token_length, prime_num_frames, tuple(patch_shape), None if text_embeds is None else tuple(text_embeds.shape),
               float(cond_scale), float(starting_temperature), float(noise_K), compact, dt, str(device), self.steps,
               self.critic_noise_anneal_schedule)
        graphs = self.__dict__.setdefault('_pk_sample_graphs', {}) if use_graph else None
        entry = graphs.get(key) if use_graph else None
        # a captured graph holds raw pointers to the live parameters AND to the packed copies derived from them: any change of a
        # parameter (optimizer step, EMA update through the autograd-visible path, load_state_dict, .to()) re-captures
        fp = param_fingerprint(self) if use_graph else None
        if entry is not None and entry['fingerprint'] != fp:
            graphs.pop(key)
            entry = None
        if entry is not None:
            st = entry['st']
            if has_ctx or c_has_ctx:
                entry['text_embeds'].copy_(text_embeds)
                entry['tm'].copy_(text_mask)
                # the CFG layouts are views/derived copies of the static inputs: rebuild them in place
                if st['ctx_r'] is not None:
                    st['ctx_r'].copy_(rep(entry['text_embeds']))
                    st['tm_r'].copy_(_cfg_masks(entry['tm'], B, with_null))
                if st['c_ctx_r'] is not None:
                    st['c_ctx_r'].copy_(crep(entry['text_embeds']))
                    st['c_tm_r'].copy_(_cfg_masks(entry['tm'], B, c_null))
            if has_prime:
                st['prime_ids'].copy_(prime_token_ids)
            st['ids'].fill_(self.mask_id)
            st['mask'].fill_(1)
            st['seed_dev'].fill_(seed_base)
            entry['graph'].replay()
            video = entry['video'] if self.__dict__.get('_pk_sample_graph_static') else entry['video'].clone()
            return (video, st['ids'].clone()) if _return_ids else video

        if use_graph and exists(text_embeds):
            text_embeds, text_mask = text_embeds.clone(), text_mask.clone()        # they become the graph's static inputs
        st = self._sample_state(B, n, prime_token_length, device, dt)
        st.update(patch_shape=patch_shape, prime_ids=prime_token_ids, prime_num_frames=prime_num_frames, ks=mask_schedule(n, self.steps),
                  cond_scale=cond_scale, starting_temperature=starting_temperature, noise_K=noise_K, compact=compact,
                  noise_fn=_noise_fn, trace=_trace, force_fn=_force_fn, seed_base=seed_base, seed_dev=None, with_null=with_null, c_null=c_null,
                  ctx_r=rep(text_embeds).contiguous() if has_ctx else None, tm_r=_cfg_masks(text_mask, B, with_null) if has_ctx else None,
                  c_ctx_r=crep(text_embeds).contiguous() if c_has_ctx else None, c_tm_r=_cfg_masks(text_mask, B, c_null) if c_has_ctx else None)
        st['ids'].fill_(self.mask_id)
        st['mask'].fill_(1)
        if not use_graph:
            video = self._sample_loop(st)
            return (video, st['ids']) if _return_ids else video

        # first call for this configuration: one eager pass (packs weights, fills the bias caches), then capture
        st['seed_dev'] = torch.zeros((1,), device=device, dtype=torch.int64)
        st['seed_dev'].fill_(seed_base)
        entry = dict(st=st, text_embeds=text_embeds, tm=text_mask, fingerprint=fp)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            self._sample_loop(st)
        torch.cuda.current_stream(device).wait_stream(side)
        st['ids'].fill_(self.mask_id)
        st['mask'].fill_(1)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            entry['video'] = self._sample_loop(st)
        entry['graph'] = graph
        graphs[key] = entry
        graph.replay()
        video = entry['video'] if self.__dict__.get('_pk_sample_graph_static') else entry['video'].clone()
        return (video, st['ids'].clone()) if _return_ids else video


def make_video(phenaki: Phenaki, texts: List[str], num_frames, prime_lengths):
    """phenaki_pytorch.py:691-714 : autoregressive scene chaining with K primed frames."""
    num_scenes = len(texts)
    num_frames = cast_tuple(num_frames, num_scenes)
    prime_lengths = cast_tuple(prime_lengths, num_scenes - 1)
    prime_lengths = (*prime_lengths, 0)       # last scene needs no priming
    video_prime = None
    scenes = []
    for text, scene_num_frames, next_scene_prime_length in zip(texts, num_frames, prime_lengths):
        video = phenaki.sample(texts=text, prime_frames=video_prime, num_frames=scene_num_frames)
        scenes.append(video)
        video_prime = video[:, :, -next_scene_prime_length:]
    return torch.cat(scenes, dim=2), scenes
