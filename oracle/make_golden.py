"""Mint golden vectors by running the REAL reference (container only).

TEST INFRASTRUCTURE ONLY.  Usage (from the repo root, needs /root/reference):

    python -m oracle.make_golden            # writes tests/golden/*.pt

The reference has no tests or known-answer vectors of its own, so these
files are the pin: the unmodified reference modules (imported via oracle/ref_shim.py),
weights from oracle/weights.py (name-keyed, reproducible), seeded inputs, and -- for the
sampler -- noise injected by replacing the reference module's ``gumbel_noise``/``uniform``
globals (phenaki_pytorch.py:69-70, 88-90) so the exact same U[0,1) draws can be fed to
the oracle and to the HIP path.  Only OUTPUTS are stored (weights/inputs are regenerated).
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import ref_shim, weights  # noqa: E402
from oracle.configs import TINY, FULL, build_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


class Recorder:
    """wraps the reference's sampling helpers to inject noise and record every step."""

    def __init__(self, R, phenaki, noise_seed_base):
        self.R, self.ph, self.base = R, phenaki, noise_seed_base
        self.steps = []
        self.cur = None
        self.scene = 0

    def install(self):
        m = self.R.module
        self._orig = (m.gumbel_noise, m.uniform, m.gumbel_sample)
        rec = self

        def gumbel_noise(t):
            u = weights.uniform_noise(tuple(t.shape), rec.base + 100 * rec.scene + 2 * rec.cur['step'])
            return -m.log(-m.log(u))

        def uniform(shape, device):
            return weights.uniform_noise(tuple(shape), rec.base + 100 * rec.scene + 2 * rec.cur['step'] + 1)

        def gumbel_sample(t, temperature=1., dim=-1):
            out = rec._orig[2](t, temperature=temperature, dim=dim)
            rec.cur['pred'] = out.clone()
            rec.cur['temperature'] = temperature
            return out

        m.gumbel_noise, m.uniform, m.gumbel_sample = gumbel_noise, uniform, gumbel_sample

        mg, cr = self.ph.maskgit, self.ph.critic
        self._mg_fwd = mg.forward_with_cond_scale

        def mg_fwd(ids, **kw):
            rec.cur = dict(step=len([s for s in rec.steps if s['scene'] == rec.scene]), scene=rec.scene)
            rec.steps.append(rec.cur)
            rec.cur['mg_input'] = ids.clone()
            out = rec._mg_fwd(ids, **kw)
            if rec.keep_logits:
                rec.cur['logits'] = out.clone()
            return out
        mg.forward_with_cond_scale = mg_fwd
        if cr is not None:
            self._cr_fwd = cr.forward_with_cond_scale

            def cr_fwd(ids, **kw):
                rec.cur['critic_input'] = ids.clone()
                out = rec._cr_fwd(ids, **kw)
                rec.cur['critic_raw'] = out.clone()
                return out
            cr.forward_with_cond_scale = cr_fwd
        return self

    def uninstall(self):
        m = self.R.module
        m.gumbel_noise, m.uniform, m.gumbel_sample = self._orig
        self.ph.maskgit.forward_with_cond_scale = self._mg_fwd
        if self.ph.critic is not None:
            self.ph.critic.forward_with_cond_scale = self._cr_fwd

    keep_logits = False


def cvivit_golden(R, cfgs, batch, frames, tag, subsample):
    cv, _, _, _ = build_reference(R, cfgs, with_phenaki=False)
    H, W = cfgs['cvivit']['image_size'], cfgs['cvivit']['image_size']
    video = weights.synthetic_video(batch, frames, H, W, seed=0)
    cap = {}
    hook = cv.vq.project_in.register_forward_hook(lambda m, i, o: cap.__setitem__('proj', o.detach().clone()))
    with torch.no_grad():
        t0 = time.time()
        ids = cv(video, return_only_codebook_ids=True)
        t_enc = time.time() - t0
        first = cv.to_patch_emb_first_frame(video[:, :, :1])
        rest = cv.to_patch_emb(video[:, :, 1:])
        patch_tokens = torch.cat((first, rest), dim=1)
        enc_tokens = cv.encode(patch_tokens)
        recon = cv.decode_from_codebook_indices(ids.flatten(1))
        recon2 = cv(video, return_recons_only=True)
    hook.remove()
    assert torch.equal(recon, recon2)
    out = dict(ids=ids, proj=cap['proj'], t_encode_ref_cpu=t_enc,
               recon_sum=recon.double().sum().item(), recon_abs=recon.double().abs().sum().item())
    if subsample:
        out['patch_tokens_sub'] = patch_tokens[:, :, ::2, ::2, ::8].clone()
        out['enc_tokens_sub'] = enc_tokens[:, :, ::2, ::2, ::8].clone()
        out['recon_sub'] = recon[:, :, ::4, ::8, ::8].clone()
    else:
        out['patch_tokens'] = patch_tokens
        out['enc_tokens'] = enc_tokens
        out['recon'] = recon
    torch.save(out, os.path.join(OUT, f'cvivit_{tag}.pt'))
    print(f'cvivit_{tag}: ids {tuple(ids.shape)} encode {t_enc:.2f}s')


def maskgit_golden(R, cfgs, batch, frames, ctx_len, tag, col_stride):
    cv, mg, cr, _ = build_reference(R, cfgs, with_phenaki=False)
    pt = cfgs['cvivit']['temporal_patch_size']
    hw = cfgs['cvivit']['image_size'] // cfgs['cvivit']['patch_size']
    patch_shape = (1 + (frames - 1) // pt, hw, hw)
    n = patch_shape[0] * hw * hw
    V = cfgs['maskgit']['num_tokens']
    g = torch.Generator().manual_seed(77)
    ids = torch.randint(0, V + 1, (batch, n), generator=g)
    ids[:, ::3] = V  # mask id
    ctx = weights.synthetic_context(batch, ctx_len, cfgs['maskgit']['dim_context'], seed=1, pad_last=3)
    text_mask = (ctx != 0).any(-1)
    with torch.no_grad():
        cond = mg(ids, video_patch_shape=patch_shape, context=ctx, text_mask=text_mask, cond_drop_prob=0.)
        null = mg(ids, video_patch_shape=patch_shape, context=ctx, text_mask=text_mask, cond_drop_prob=1.)
        cfg = mg.forward_with_cond_scale(ids, video_patch_shape=patch_shape, context=ctx, text_mask=text_mask, cond_scale=5.)
        embeds = mg(ids, video_patch_shape=patch_shape, context=ctx, text_mask=text_mask, return_embeds=True)
        bias = mg.continuous_pos_bias(*patch_shape)
        sc = cr.forward_with_cond_scale(ids, video_patch_shape=patch_shape, context=ctx, text_mask=text_mask, cond_scale=5.)
        sc_cond = cr(ids, video_patch_shape=patch_shape, context=ctx, text_mask=text_mask, cond_drop_prob=0.)
    out = dict(ids=ids, patch_shape=patch_shape, ctx_len=ctx_len,
               cond=cond[:, :, ::col_stride].clone(), null=null[:, :, ::col_stride].clone(),
               cfg=cfg[:, :, ::col_stride].clone(), cfg_argmax=cfg.argmax(-1), cfg_lse=cfg.logsumexp(-1),
               embeds=embeds[:, :, ::max(1, col_stride // 64)].clone(), bias_sub=bias[:, ::7, ::5].clone(),
               critic_cfg=sc, critic_cond=sc_cond, col_stride=col_stride)
    torch.save(out, os.path.join(OUT, f'maskgit_{tag}.pt'))
    print(f'maskgit_{tag}: logits {tuple(cfg.shape)}')


def sample_golden(R, cfgs, batch, frames_list, prime_len, ctx_len, tag, keep_logits, with_critic=True, steps=None):
    cv, mg, cr, ph = build_reference(R, cfgs, with_phenaki=True, with_critic=with_critic, steps=steps)
    ctx = weights.synthetic_context(batch, ctx_len, cfgs['maskgit']['dim_context'], seed=2)
    ph.encode_texts = lambda texts, output_device=None: ctx
    rec = Recorder(R, ph, noise_seed_base=500)
    rec.keep_logits = keep_logits
    rec.install()
    t0 = time.time()
    try:
        with torch.no_grad():
            videos = []
            prime = None
            for si, nf in enumerate(frames_list):
                rec.scene = si
                v = ph.sample(texts=['x'] * batch, num_frames=nf, prime_frames=prime, cond_scale=5.)
                videos.append(v)
                prime = v[:, :, -prime_len:] if prime_len else None
    finally:
        rec.uninstall()
    dt = time.time() - t0
    out = dict(steps=rec.steps, frames_list=frames_list, prime_len=prime_len, ctx_len=ctx_len, batch=batch,
               t_sample_ref_cpu=dt, with_critic=with_critic)
    if keep_logits:
        out['videos'] = videos
    else:
        out['videos_sub'] = [v[:, :, ::4, ::8, ::8].clone() for v in videos]
        out['videos_sum'] = [v.double().sum().item() for v in videos]
    torch.save(out, os.path.join(OUT, f'sample_{tag}.pt'))
    print(f'sample_{tag}: {len(rec.steps)} step records, {dt:.1f}s')


def make_video_golden(R, cfgs, frames, prime_lengths, ctx_lens, tag, keep_videos, steps=None):
    """the reference's OWN make_video (phenaki_pytorch.py:691-714) -- one text per scene, each scene primed by the last K frames of the
    previous one -- with the noise injected per (scene, step) as in sample_golden.  Every scene gets its own text context (different lengths).
    Batch is 1 (make_video hands `sample` a single str).  Recorded per step: the MaskGit input ids (prime tokens included), the predicted ids,
    the critic input; per scene the video (whole, or sub-sampled + checksums at full size)."""
    cv, mg, cr, ph = build_reference(R, cfgs, with_phenaki=True, with_critic=True, steps=steps)
    texts = [f'scene {i}' for i in range(len(frames))]
    ctxs = {t: weights.synthetic_context(1, L, cfgs['maskgit']['dim_context'], seed=20 + i) for i, (t, L) in enumerate(zip(texts, ctx_lens))}
    ph.encode_texts = lambda tx, output_device=None: torch.cat([ctxs[t] for t in tx], 0)
    rec = Recorder(R, ph, noise_seed_base=500)
    rec.keep_logits = False
    rec.install()
    orig_sample = ph.sample
    calls = []

    def sample(**kw):
        rec.scene = len(calls)
        calls.append(dict(num_frames=kw['num_frames'], prime_frames=None if kw.get('prime_frames') is None else kw['prime_frames'].shape[2],
                          text=kw['texts']))
        return orig_sample(**kw)
    ph.sample = sample
    t0 = time.time()
    try:
        with torch.no_grad():
            whole, scenes = R.module.make_video(ph, texts, frames, prime_lengths)
    finally:
        rec.uninstall()
        del ph.sample
    dt = time.time() - t0
    out = dict(steps=rec.steps, frames=tuple(frames), prime_lengths=prime_lengths, ctx_lens=tuple(ctx_lens), texts=texts, calls=calls,
               whole_shape=tuple(whole.shape), t_ref_cpu=dt)
    if keep_videos:
        out['scenes'] = scenes
    else:
        out['scenes_sub'] = [v[:, :, ::3, ::8, ::8].clone() for v in scenes]
        out['scenes_sum'] = [v.double().sum().item() for v in scenes]
        out['scenes_abs'] = [v.double().abs().sum().item() for v in scenes]
    torch.save(out, os.path.join(OUT, f'make_video_{tag}.pt'))
    print(f'make_video_{tag}: {len(rec.steps)} step records over {len(calls)} scenes, whole {tuple(whole.shape)}, {dt:.1f}s')


def sample_ragged_golden(R, cfgs, batch, frames, ctx_lens, tag):
    """Phenaki.sample at batch > 1 with captions of DIFFERENT lengths: t5_encode_text zero-fills the pads (t5.py:94-103), the per-row
    text_mask is any(embeds != 0) (phenaki_pytorch.py:455-463).  cond_scale 5, TokenCritic, full step count; ids per step + sub-sampled video."""
    cv, mg, cr, ph = build_reference(R, cfgs, with_phenaki=True, with_critic=True)
    ctx = weights.ragged_context(ctx_lens, cfgs['maskgit']['dim_context'], seed=6)
    assert ctx.shape[0] == batch
    ph.encode_texts = lambda texts, output_device=None: ctx
    rec = Recorder(R, ph, noise_seed_base=700)
    rec.keep_logits = False
    rec.install()
    t0 = time.time()
    try:
        with torch.no_grad():
            v = ph.sample(texts=['x'] * batch, num_frames=frames, cond_scale=5.)
    finally:
        rec.uninstall()
    dt = time.time() - t0
    out = dict(steps=rec.steps, frames=frames, ctx_lens=tuple(ctx_lens), batch=batch, t_sample_ref_cpu=dt,
               video_sub=v[:, :, ::4, ::8, ::8].clone(), video_sum=v.double().sum().item(), video_abs=v.double().abs().sum().item())
    torch.save(out, os.path.join(OUT, f'sample_{tag}.pt'))
    print(f'sample_{tag}: {len(rec.steps)} step records, {dt:.1f}s')



def forward_golden(R, cfgs, batch, frames, ctx_len, tag):
    """Phenaki.forward (the training objective, value only) of the real reference with its three random draws replaced
    by deterministic ones (torch.randint / torch.rand patched for the duration of the call, gumbel_noise as in the
    sampler goldens), so the oracle and the HIP path can be fed the same draws."""
    cv, mg, cr, ph = build_reference(R, cfgs, with_phenaki=True, with_critic=True)
    m = R.module
    H = cfgs['cvivit']['image_size']
    video = weights.synthetic_video(batch, frames, H, H, seed=5)
    ctx = weights.synthetic_context(batch, ctx_len, cfgs['maskgit']['dim_context'], seed=3, pad_last=2)
    steps = cfgs['steps']
    rand_step = (torch.arange(batch) * 2 + 1) % steps
    rec = {}
    orig = (torch.randint, torch.rand, m.gumbel_noise, m.gumbel_sample)

    def randint(low, high, size, **kw):
        assert (low, high, tuple(size)) == (0, steps, (batch,))
        return rand_step.clone()

    pt = cfgs['cvivit']['temporal_patch_size']
    hw = H // cfgs['cvivit']['patch_size']
    n = (1 + (frames - 1) // pt) * hw * hw
    perm_noise = weights.uniform_noise((batch, n), 700)                 # drawn BEFORE torch.rand is patched
    gumbel_u = weights.uniform_noise((batch, n, cfgs['maskgit']['num_tokens']), 701)

    def rand(size, **kw):
        assert tuple(size) == (batch, n)
        return perm_noise.clone()

    def gumbel_noise(t):
        assert tuple(t.shape) == tuple(gumbel_u.shape)
        return -m.log(-m.log(gumbel_u))

    def gumbel_sample(t, temperature=1., dim=-1):
        out = orig[3](t, temperature=temperature, dim=dim)
        rec['pred'] = out.clone()
        return out

    torch.randint, torch.rand, m.gumbel_noise, m.gumbel_sample = randint, rand, gumbel_noise, gumbel_sample
    try:
        with torch.no_grad():
            ids = cv(video, return_only_codebook_ids=True)
            total = ph(video, text_embeds=ctx)
            gen = ph(video_codebook_ids=ids, text_embeds=ctx, only_train_generator=True)
            crit = ph(video_codebook_ids=ids, text_embeds=ctx, only_train_critic=True)
    finally:
        torch.randint, torch.rand, m.gumbel_noise, m.gumbel_sample = orig
    out = dict(ids=ids, loss=total, loss_generator=gen, loss_critic=crit, pred=rec['pred'], rand_step=rand_step,
               batch=batch, frames=frames, ctx_len=ctx_len, critic_loss_weight=ph.critic_loss_weight,
               critic_temperature=ph.critic_train_sample_temperature)
    torch.save(out, os.path.join(OUT, f'forward_{tag}.pt'))
    print(f'forward_{tag}: loss {float(total):.6f} = generator {float(gen):.6f} + w * critic {float(crit):.6f}')


def forward_grads_golden(R, cfgs, batch, frames, ctx_len, tag):
    """the REAL reference's training step -- Phenaki.forward (phenaki_pytorch.py:562-687) + loss.backward() -- with the
    three random draws of forward_golden: the gradient of every MaskGit and TokenCritic parameter (total objective), plus the
    generator-only / critic-only variants' losses.  Token ids come from the reference C-ViViT (frozen, no grad, as in :580-584)."""
    cv, mg, cr, ph = build_reference(R, cfgs, with_phenaki=True, with_critic=True)
    m = R.module
    H = cfgs['cvivit']['image_size']
    video = weights.synthetic_video(batch, frames, H, H, seed=5)
    ctx = weights.synthetic_context(batch, ctx_len, cfgs['maskgit']['dim_context'], seed=3, pad_last=2)
    steps = cfgs['steps']
    rand_step = (torch.arange(batch) * 2 + 1) % steps
    orig = (torch.randint, torch.rand, m.gumbel_noise)
    pt = cfgs['cvivit']['temporal_patch_size']
    hw = H // cfgs['cvivit']['patch_size']
    n = (1 + (frames - 1) // pt) * hw * hw
    perm_noise = weights.uniform_noise((batch, n), 700)
    gumbel_u = weights.uniform_noise((batch, n, cfgs['maskgit']['num_tokens']), 701)

    def randint(low, high, size, **kw):
        return rand_step.clone()

    def rand(size, **kw):
        return perm_noise.clone()

    def gumbel_noise(t):
        return -m.log(-m.log(gumbel_u))

    with torch.no_grad():
        ids = cv(video, return_only_codebook_ids=True)
    out = dict(ids=ids, rand_step=rand_step, batch=batch, frames=frames, ctx_len=ctx_len)
    torch.randint, torch.rand, m.gumbel_noise = randint, rand, gumbel_noise
    try:
        for name, kw in (('total', {}), ('generator', dict(only_train_generator=True)), ('critic', dict(only_train_critic=True))):
            for mod in (mg, cr):
                mod.train()
                mod.zero_grad(set_to_none=True)
            for prm in list(mg.parameters()) + list(cr.parameters()):
                prm.requires_grad_(True)
            loss = ph(video_codebook_ids=ids, text_embeds=ctx, **kw)
            loss.backward()
            out[f'loss_{name}'] = loss.detach().clone()
            grads = {f'maskgit.{k}': v.grad.detach().clone() for k, v in mg.named_parameters() if v.grad is not None}
            grads.update({f'critic.{k}': v.grad.detach().clone() for k, v in cr.named_parameters() if v.grad is not None})
            if name == 'total':
                out['grads_total'] = grads
            else:
                # the variants train one of the two networks: which parameters received a gradient, and one probe value each
                out[f'grad_keys_{name}'] = sorted(grads)
                probe = 'maskgit.to_logits.weight' if name == 'generator' else 'critic.to_logits.0.weight'
                out[f'grad_probe_{name}'] = (probe, grads[probe])
    finally:
        torch.randint, torch.rand, m.gumbel_noise = orig
    torch.save(out, os.path.join(OUT, f'forward_grads_{tag}.pt'))
    print(f'forward_grads_{tag}: losses {float(out["loss_total"]):.6f} / {float(out["loss_generator"]):.6f} / {float(out["loss_critic"]):.6f}; '
          f'{len(out["grads_total"])} gradients ({len(out["grad_keys_generator"])} generator-only, {len(out["grad_keys_critic"])} critic-only)')


def recon_loss_golden(R, cfgs, tag):
    """CViViT.forward's default return with use_vgg_and_gan=False: the reconstruction MSE (cvivit.py:585-627), plain, with a
    frame mask, with return_recons, and for a 4-D image batch."""
    cv, _, _, _ = build_reference(R, cfgs, with_phenaki=False)
    H = cfgs['cvivit']['image_size']
    video = weights.synthetic_video(2, 5, H, H, seed=6)
    mask = torch.tensor([[True] * 5, [True] * 3 + [False] * 2])
    with torch.no_grad():
        loss = cv(video)
        loss_masked = cv(video, mask=mask)
        loss_r, recon = cv(video, return_recons=True)
        loss_img = cv(video[:, :, 0])
    assert float(loss_r) == float(loss)
    out = dict(loss=loss.clone(), loss_masked=loss_masked.clone(), loss_image=loss_img.clone(), mask=mask, recon_sum=recon.double().sum().item())
    torch.save(out, os.path.join(OUT, f'recon_loss_{tag}.pt'))
    print(f'recon_loss_{tag}: {float(loss):.6f} masked {float(loss_masked):.6f} image {float(loss_img):.6f}')


def cvivit_grads_golden(R, cfgs, tag):
    """the REAL reference's tokenizer training step -- CViViT(use_vgg_and_gan=False).train()(video) (cvivit.py:518-627:
    the reconstruction MSE through the straight-through LFQ of oracle/lfq.py) + loss.backward(): loss and the gradient of every parameter,
    for a video batch, for a 4-D image batch and for the video batch under a frame mask (variable-length training, :585-589)."""
    cv, _, _, _ = build_reference(R, cfgs, with_phenaki=False)
    H = cfgs['cvivit']['image_size']
    video = weights.synthetic_video(2, 5, H, H, seed=8)
    out = {}
    mask = torch.tensor([[True] * 5, [True] * 3 + [False] * 2])
    out['mask'] = mask
    for name, x in (('video', video), ('image', video[:, :, 2]), ('masked', video)):
        cv.train()
        cv.zero_grad(set_to_none=True)
        loss = cv(x, mask=mask) if name == 'masked' else cv(x)
        loss.backward()
        out[f'loss_{name}'] = loss.detach().clone()
        # per parameter: the 2-norm of the whole gradient (f64) and every stride-th element of it (<= 4096 values): the fixture stays small
        grads = {}
        for k, v in cv.named_parameters():
            if v.grad is None:
                continue
            flat = v.grad.detach().reshape(-1)
            stride = max(1, -(-flat.numel() // 4096))
            grads[k] = dict(norm=float(flat.double().norm()), stride=stride, sample=flat[::stride].clone(), shape=tuple(v.shape))
        out[f'grads_{name}'] = grads
    cv.eval()
    torch.save(out, os.path.join(OUT, f'cvivit_grads_{tag}.pt'))
    print(f'cvivit_grads_{tag}: loss {float(out["loss_video"]):.6f} ({len(out["grads_video"])} gradients), image {float(out["loss_image"]):.6f} '
          f'({len(out["grads_image"])} gradients)')


def selfcritic_golden(R, cfgs, tag):
    """Phenaki(self_token_critic=True) of the real reference (phenaki_pytorch.py:306-336, 374-375): the SelfCritic scores (plain and
    with classifier-free guidance) on fixed ids, and a full free-running sample whose score step is the self critic."""
    cv, mg, _, _ = build_reference(R, cfgs, with_phenaki=False, with_critic=False)
    ph = R.Phenaki(maskgit=mg, cvivit=cv, self_token_critic=True, steps=cfgs['steps'], text_embed_dim=cfgs['maskgit']['dim_context'])
    ph.eval()
    weights.fill_module(ph.critic.to_pred, salt=4)
    pt = cfgs['cvivit']['temporal_patch_size']
    hw = cfgs['cvivit']['image_size'] // cfgs['cvivit']['patch_size']
    batch, frames, ctx_len = 2, 5, 7
    patch_shape = (1 + (frames - 1) // pt, hw, hw)
    n = patch_shape[0] * hw * hw
    V = cfgs['maskgit']['num_tokens']
    g = torch.Generator().manual_seed(79)
    ids = torch.randint(0, V + 1, (batch, n), generator=g)
    ids[:, ::4] = V
    ctx = weights.synthetic_context(batch, ctx_len, cfgs['maskgit']['dim_context'], seed=1, pad_last=3)
    text_mask = (ctx != 0).any(-1)
    kw = dict(video_patch_shape=patch_shape, context=ctx, text_mask=text_mask)
    with torch.no_grad():
        sc_cfg = ph.critic.forward_with_cond_scale(ids, cond_scale=5., **kw)
        sc_cond = ph.critic(ids, cond_drop_prob=0., **kw)
    sctx = weights.synthetic_context(batch, 6, cfgs['maskgit']['dim_context'], seed=2)
    ph.encode_texts = lambda texts, output_device=None: sctx
    rec = Recorder(R, ph, noise_seed_base=500)
    rec.keep_logits = False
    rec.install()
    try:
        with torch.no_grad():
            video = ph.sample(texts=['x'] * batch, num_frames=frames, cond_scale=5.)
    finally:
        rec.uninstall()
    out = dict(ids=ids, patch_shape=patch_shape, ctx_len=ctx_len, critic_cfg=sc_cfg, critic_cond=sc_cond,
               steps=rec.steps, video=video, batch=batch, frames=frames, sample_ctx_len=6)
    torch.save(out, os.path.join(OUT, f'selfcritic_{tag}.pt'))
    print(f'selfcritic_{tag}: scores {tuple(sc_cfg.shape)}, {len(rec.steps)} sampling steps')


def unconditional_golden(R, cfgs, tag):
    """an unconditional MaskGit (phenaki_pytorch.py:125-147: no cross-attention) with a TokenCritic without cross-attention: logits,
    critic scores and a full free-running sample (no texts) of the real reference."""
    cv = R.CViViT(use_vgg_and_gan=False, **cfgs['cvivit'])
    mg = R.MaskGit(**{**cfgs['maskgit'], 'unconditional': True})
    cr = R.TokenCritic(**{**cfgs['critic'], 'has_cross_attn': False})
    weights.fill_module(cv, salt=1); weights.fill_module(mg, salt=2); weights.fill_module(cr, salt=3)
    cv.eval(); mg.eval(); cr.eval()
    ph = R.Phenaki(maskgit=mg, cvivit=cv, critic=cr, steps=cfgs['steps'], text_embed_dim=cfgs['maskgit']['dim_context'])
    ph.eval()
    pt = cfgs['cvivit']['temporal_patch_size']
    hw = cfgs['cvivit']['image_size'] // cfgs['cvivit']['patch_size']
    batch, frames = 2, 5
    patch_shape = (1 + (frames - 1) // pt, hw, hw)
    n = patch_shape[0] * hw * hw
    V = cfgs['maskgit']['num_tokens']
    g = torch.Generator().manual_seed(80)
    ids = torch.randint(0, V + 1, (batch, n), generator=g)
    ids[:, 1::3] = V
    with torch.no_grad():
        logits = mg(ids, video_patch_shape=patch_shape)
        scores = cr(ids, video_patch_shape=patch_shape)
    rec = Recorder(R, ph, noise_seed_base=500)
    rec.keep_logits = False
    rec.install()
    try:
        with torch.no_grad():
            video = ph.sample(num_frames=frames, batch_size=batch)
    finally:
        rec.uninstall()
    out = dict(ids=ids, patch_shape=patch_shape, logits=logits, critic=scores, steps=rec.steps, video=video, batch=batch, frames=frames)
    torch.save(out, os.path.join(OUT, f'unconditional_{tag}.pt'))
    print(f'unconditional_{tag}: logits {tuple(logits.shape)}, {len(rec.steps)} sampling steps')


def ce_grad_golden(R, cfgs, tag):
    """gradients of the masked-token cross entropy (phenaki_pytorch.py:640-643) of the REAL reference under autograd, at the vocabulary
    head: the to_logits input rows of the masked positions, their targets, d loss / d rows, d loss / d to_logits.weight, .bias.
    `Phenaki.forward(only_train_generator=True)` with the same patched draws as forward_golden; hooks only, nothing is modified."""
    cv, mg, cr, ph = build_reference(R, cfgs, with_phenaki=True, with_critic=True)
    m = R.module
    batch, frames, ctx_len = 3, 5, 6
    H = cfgs['cvivit']['image_size']
    steps = cfgs['steps']
    pt = cfgs['cvivit']['temporal_patch_size']
    hw = H // cfgs['cvivit']['patch_size']
    n = (1 + (frames - 1) // pt) * hw * hw
    g = torch.Generator().manual_seed(81)
    ids = torch.randint(0, cfgs['maskgit']['num_tokens'], (batch, 1 + (frames - 1) // pt, hw, hw), generator=g)
    ctx = weights.synthetic_context(batch, ctx_len, cfgs['maskgit']['dim_context'], seed=3, pad_last=2)
    rand_step = (torch.arange(batch) * 2 + 1) % steps
    perm_noise = weights.uniform_noise((batch, n), 700)
    rec = {}
    orig = (torch.randint, torch.rand, m.get_mask_subset_with_prob)

    def randint(low, high, size, **kw):
        return rand_step.clone()

    def rand(size, **kw):
        return perm_noise.clone()

    def mask_subset(mask, prob):
        out = orig[2](mask, prob)
        rec['mask'] = out.clone()
        return out

    def fwd_hook(mod, inp, out):
        rec['x'] = inp[0].detach().clone()
        inp[0].register_hook(lambda gr: rec.__setitem__('dx', gr.detach().clone()))

    h = mg.to_logits.register_forward_hook(fwd_hook)
    torch.randint, torch.rand, m.get_mask_subset_with_prob = randint, rand, mask_subset
    try:
        for p_ in ph.parameters():
            p_.grad = None
        loss = ph(video_codebook_ids=ids, text_embeds=ctx, only_train_generator=True)
        loss.backward()
    finally:
        torch.randint, torch.rand, m.get_mask_subset_with_prob = orig
        h.remove()
    mask = rec['mask']
    out = dict(loss=loss.detach().clone(), rows=rec['x'][mask].clone(), targets=ids.flatten(1)[mask].clone(), d_rows=rec['dx'][mask].clone(),
               d_rows_unmasked_absmax=rec['dx'][~mask].abs().max().item(), d_weight=mg.to_logits.weight.grad.detach().clone(),
               d_bias=mg.to_logits.bias.grad.detach().clone(), num_rows=int(mask.sum()))
    torch.save(out, os.path.join(OUT, f'ce_grad_{tag}.pt'))
    print(f"ce_grad_{tag}: loss {float(loss):.6f}, {out['num_rows']} masked rows, |dx| outside the mask {out['d_rows_unmasked_absmax']:.1e}")


def t5_golden(tag, cfg, B, L, sub):
    """the REAL HuggingFace T5EncoderModel (what the reference's t5.py:64-103 runs) on name-keyed random weights: last_hidden_state with the
    pads zero-filled as in t5.py:97-100.  No checkpoint exists offline; the architecture and arithmetic are what is pinned."""
    from transformers import T5Config, T5EncoderModel
    from oracle import t5_oracle as T
    hf = T5EncoderModel(T5Config(**cfg)).eval()
    weights.fill_module(hf, salt=5)
    ids, mask = T.t5_inputs(cfg, B, L)
    with torch.no_grad():
        out = hf(input_ids=ids, attention_mask=mask.long()).last_hidden_state
    out = out.masked_fill(~mask[..., None], 0.)
    keys = {k: list(v.shape) for k, v in hf.state_dict().items()}
    torch.save(dict(ids=ids, mask=mask, out=out[:, :, ::sub].clone(), sub=sub, keys=keys, out_absmax=out.abs().max().item()),
               os.path.join(OUT, f't5_{tag}.pt'))
    print(f't5_{tag}: out {tuple(out.shape)} absmax {out.abs().max().item():.3f}')


def _grad_summary(module):
    """per parameter: the 2-norm of the whole gradient (f64) and every stride-th element of it (<= 4096 values): the fixture stays small"""
    grads = {}
    for k, v in module.named_parameters():
        if v.grad is None or k.startswith('vgg.'):
            continue
        flat = v.grad.detach().reshape(-1)
        stride = max(1, -(-flat.numel() // 4096))
        grads[k] = dict(norm=float(flat.double().norm()), stride=stride, sample=flat[::stride].clone(), shape=tuple(v.shape))
    return grads


def gan_golden(R, cfgs, tag, batch=2, frames=5):
    """SURVEY.md 8f row 4, the adversarial half: the REAL reference's CViViT(use_vgg_and_gan=True, vgg=<stub>) in training mode --
    (1) Discriminator logits on a fixed image batch; (2) forward(video, return_discr_loss=True) (cvivit.py:604-622: hinge + gradient penalty)
    + backward: loss and every discriminator gradient; (3) forward(video) (cvivit.py:585-671: recon + perceptual + adaptive_weight * gen) +
    backward: loss, its parts and every gradient; (4) the same under a frame mask.  The frame the reference picks (torch.randn on the host
    generator after torch.manual_seed) is stored, so the oracle can be fed the same choice."""
    H = cfgs['cvivit']['image_size']
    vgg = weights.stub_vgg(H)
    cv = R.CViViT(use_vgg_and_gan=True, vgg=vgg, **cfgs['cvivit'])
    weights.fill_module(cv, salt=1)
    cv.train()
    video = weights.synthetic_video(batch, frames, H, H, seed=12)
    mask = torch.tensor([[True] * frames, [True] * (frames - 2) + [False] * 2][:batch])
    out = dict(mask=mask, discr_keys={k: list(v.shape) for k, v in cv.state_dict().items() if k.startswith('discr.')})
    imgs = weights.synthetic_video(3, 1, H, H, seed=13)[:, :, 0]
    with torch.no_grad():
        out['logits'] = cv.discr(imgs).clone()

    def frames_for(seed, m=None):
        torch.manual_seed(seed)
        logits = torch.randn(batch, frames)
        if m is not None:
            logits = logits.masked_fill(~m, -torch.finfo(logits.dtype).max)
        return logits.topk(1, dim=-1).indices.reshape(-1)

    # (2) discriminator step
    out['frame_discr'] = frames_for(21)
    cv.zero_grad(set_to_none=True)
    torch.manual_seed(21)
    loss = cv(video, return_discr_loss=True)
    loss.backward()
    out['loss_discr'] = loss.detach().clone()
    out['grads_discr'] = {k: v for k, v in _grad_summary(cv).items() if k.startswith('discr.')}
    # the two terms separately (the reference returns only their sum): hinge alone = the same call with the penalty's weight removed
    with torch.no_grad():
        recon = cv(video, return_recons_only=True)
    real = R.cvivit.pick_video_frame(video, out['frame_discr'][:, None])
    fake = R.cvivit.pick_video_frame(recon, out['frame_discr'][:, None])
    with torch.no_grad():
        out['hinge_discr'] = R.cvivit.hinge_discr_loss(cv.discr(fake), cv.discr(real)).clone()
    # (3) generator step, (4) with a frame mask
    # the adaptive weight (cvivit.py:657-664) is the one scalar that scales every tokenizer gradient of the generator term: record its two gradient
    # norms by wrapping the reference module's OWN safe_div (a module global; restored below) -- the parity tests pin it separately
    seen = {}
    orig_safe_div = R.cvivit.safe_div

    def recording_safe_div(numer, denom, eps=1e-8):
        seen['n_per'], seen['n_gen'] = numer.detach().clone(), denom.detach().clone()
        return orig_safe_div(numer, denom, eps)
    R.cvivit.safe_div = recording_safe_div
    # ... and the two component gradients it is the ratio of (grad_layer_wrt_loss, cvivit.py:97-103; called for gen_loss first, then perceptual_loss)
    orig_glwl = R.cvivit.grad_layer_wrt_loss
    comp = []

    def recording_glwl(loss, layer):
        g = orig_glwl(loss, layer)
        comp.append(g.detach().clone())
        return g
    R.cvivit.grad_layer_wrt_loss = recording_glwl
    # vector_quantize_pytorch is absent, so the quantizer inside this "reference" run is oracle/lfq.py's restatement (ref_shim).  Its sign /
    # straight-through part is elementary; its AUXILIARY loss (entropy + commitment, defaults written from the published code) is not pinned by any upstream
    # vector.  The generator step is therefore minted twice: 'gen' / 'gen_masked' with the aux term (SELF-DERIVED for that term: the test says so and holds
    # it to 5e-4), and 'gen_noaux' with both aux weights set to 0 -- an objective that does not depend on the restated aux formula at all (held to 2e-4).
    from oracle import lfq as _lfq
    out['lfq_aux'] = dict(self_derived=True, source='oracle/lfq.py (restated; vector-quantize-pytorch absent, setup.py pins >=1.11.8 only)',
                          defaults=dict(_lfq.LFQ_DEFAULTS), log_eps=_lfq.LOG_EPS)
    w_aux = (cv.vq.entropy_loss_weight, cv.vq.commitment_loss_weight)
    for name, m, seed in (('gen', None, 22), ('gen_masked', mask, 23), ('gen_noaux', None, 22)):
        cv.vq.entropy_loss_weight, cv.vq.commitment_loss_weight = (0., 0.) if name == 'gen_noaux' else w_aux
        out[f'frame_{name}'] = frames_for(seed, m)
        cv.zero_grad(set_to_none=True)
        torch.manual_seed(seed)
        comp.clear()
        loss = cv(video, mask=m) if m is not None else cv(video)
        loss.backward()
        out[f'parts_{name}'] = dict(norm_grad_perceptual=seen['n_per'], norm_grad_gen=seen['n_gen'],
                                    adaptive_weight=(seen['n_per'] / (seen['n_gen'] + 1e-8)).clamp(max=1e4),
                                    grad_gen=comp[0].reshape(-1)[::7].clone(), grad_perceptual=comp[1].reshape(-1)[::7].clone())     # strided samples of the (P, dim) gradients
        out[f'loss_{name}'] = loss.detach().clone()
        out[f'grads_{name}'] = _grad_summary(cv)
    cv.vq.entropy_loss_weight, cv.vq.commitment_loss_weight = w_aux
    R.cvivit.safe_div = orig_safe_div
    R.cvivit.grad_layer_wrt_loss = orig_glwl
    torch.save(out, os.path.join(OUT, f'gan_{tag}.pt'))
    print(f'gan_{tag}: discr loss {float(out["loss_discr"]):.6f} (hinge {float(out["hinge_discr"]):.6f}, {len(out["grads_discr"])} gradients), '
          f'generator loss {float(out["loss_gen"]):.6f} ({len(out["grads_gen"])} gradients), masked {float(out["loss_gen_masked"]):.6f}; '
          f'frames {out["frame_discr"].tolist()} {out["frame_gen"].tolist()} {out["frame_gen_masked"].tolist()}')


def keys_golden(R):
    """state_dict contract: every key, shape and dtype of the reference modules."""
    import json
    out = {}
    for tag, cfgs in (('tiny', TINY), ('full', FULL)):
        cv, mg, cr, _ = build_reference(R, cfgs, with_phenaki=False)
        for kind, m in (('cvivit', cv), ('maskgit', mg), ('critic', cr)):
            out[f'{tag}.{kind}'] = {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()}
    with open(os.path.join(OUT, 'state_dict_keys.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print('state_dict_keys.json written')


def main():
    os.makedirs(OUT, exist_ok=True)
    R = ref_shim.load()
    torch.set_num_threads(os.cpu_count())
    which = sys.argv[1:] or ['tiny', 'full']
    if 'keys' in which or 'tiny' in which:
        keys_golden(R)
    if 'tiny' in which:
        cvivit_golden(R, TINY, batch=2, frames=5, tag='tiny', subsample=False)
        maskgit_golden(R, TINY, batch=2, frames=5, ctx_len=7, tag='tiny', col_stride=1)
        sample_golden(R, TINY, batch=2, frames_list=[5], prime_len=0, ctx_len=6, tag='tiny', keep_logits=True)
        sample_golden(R, TINY, batch=2, frames_list=[5], prime_len=0, ctx_len=6, tag='tiny_nocritic', keep_logits=True,
                      with_critic=False)
        sample_golden(R, TINY, batch=1, frames_list=[5, 4], prime_len=3, ctx_len=5, tag='tiny_primed', keep_logits=True)
    if 'tiny' in which or 'forward' in which:
        forward_golden(R, TINY, batch=3, frames=5, ctx_len=6, tag='tiny')
        recon_loss_golden(R, TINY, tag='tiny')
    if 'tiny' in which or 't5' in which:
        from oracle import t5_oracle as T
        t5_golden('tiny', T.T5_TINY, B=3, L=21, sub=1)
        t5_golden('base', T.T5_BASE, B=2, L=40, sub=8)
    if 'tiny' in which or 'grads' in which:
        ce_grad_golden(R, TINY, tag='tiny')
    if 'tiny' in which or 'grads' in which or 'fwdgrads' in which:
        forward_grads_golden(R, TINY, batch=3, frames=5, ctx_len=6, tag='tiny')
    if 'tiny' in which or 'grads' in which or 'cvgrads' in which:
        cvivit_grads_golden(R, TINY, tag='tiny')
    if 'tiny' in which or 'gan' in which:
        gan_golden(R, TINY, tag='tiny')
    if 'tiny' in which or 'critics' in which:
        selfcritic_golden(R, TINY, tag='tiny')
        unconditional_golden(R, TINY, tag='tiny')
    if 'tiny' in which or 'make_video' in which:
        make_video_golden(R, TINY, frames=(5, 4, 4), prime_lengths=3, ctx_lens=(5, 7, 4), tag='tiny', keep_videos=True)
        make_video_golden(R, TINY, frames=(5, 4, 6), prime_lengths=(3, 1), ctx_lens=(6, 3, 8), tag='tiny_perscene', keep_videos=True)
        sample_ragged_golden(R, TINY, batch=3, frames=5, ctx_lens=(6, 3, 5), tag='tiny_ragged')
    if 'full' in which or 'make_video_full' in which:
        # BASELINE configs[4]: 3 scenes (17, 14, 14 frames), K = 5 -> scenes 2 / 3 run n = 192 prime + 448 new tokens, patch shape (10, 8, 8)
        make_video_golden(R, FULL, frames=(17, 14, 14), prime_lengths=5, ctx_lens=(12, 9, 14), tag='full', keep_videos=False)
    if 'full' in which or 'ragged_full' in which:
        # BASELINE configs[3]'s per-GPU share: B = 4, CFG 5, two caption lengths (zero-filled pads -> per-row text_mask)
        sample_ragged_golden(R, FULL, batch=4, frames=17, ctx_lens=(12, 7, 12, 7), tag='full_b4_ragged')
    if 'full' in which:
        cvivit_golden(R, FULL, batch=2, frames=17, tag='full', subsample=True)
        maskgit_golden(R, FULL, batch=1, frames=17, ctx_len=12, tag='full', col_stride=512)
        sample_golden(R, FULL, batch=1, frames_list=[17], prime_len=0, ctx_len=12, tag='full', keep_logits=False)


if __name__ == '__main__':
    main()
