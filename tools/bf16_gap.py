"""How close is the bf16 HIP path to the oracle run at the same precision (oracle.precision('bf16')), compared with the plain
bf16-vs-f32 gap?  Prints max-norm and RMS relative errors per quantity and per depth.  GPU box: python tools/bf16_gap.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_grad_enabled(False)
from oracle import phenaki_oracle as O, weights, hostcpu            # noqa: E402
from oracle.configs import FULL, oracle_cfgs, state_dicts          # noqa: E402
import phenaki_pytorch_amd as P                                      # noqa: E402

hostcpu.configure()


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    d = (a - b)
    return d.abs().max().item() / b.abs().max().item(), (d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


def show(what, gpu, ob, of):
    (m1, r1), (m2, r2), (m3, r3) = rel(gpu, ob), rel(gpu, of), rel(ob, of)
    print(f'{what:44s} gpu-vs-oracle_bf16 max {m1:.2e} rms {r1:.2e} | gpu-vs-oracle_f32 max {m2:.2e} rms {r2:.2e} | oracle_bf16-vs-f32 max {m3:.2e} rms {r3:.2e}', flush=True)


cv_sd, mg_sd, cr_sd = state_dicts('full')
video = weights.synthetic_video(2, 17, 256, 256, seed=0)
for sdepth, tdepth in ((1, 0), (1, 1), (2, 2), (4, 4)):
    cfg = {**FULL, 'cvivit': {**FULL['cvivit'], 'spatial_depth': sdepth, 'temporal_depth': tdepth}}
    cvc, _, _ = oracle_cfgs(cfg)
    cv = P.CViViT(use_vgg_and_gan=False, **cfg['cvivit'])
    cv.load_state_dict(cv_sd, strict=False)
    cv = cv.cuda().eval()
    P.set_compute_dtype(cv, 'bf16')
    tok_f = O.cvivit_patch_embed(cv_sd, cvc, video)
    with O.precision('bf16'):
        tok_b = O.cvivit_patch_embed(cv_sd, cvc, video)
        enc_b = O.cvivit_encode(cv_sd, cvc, tok_b)
    enc_f = O.cvivit_encode(cv_sd, cvc, tok_b)
    tok_g, T = cv._patch_embed(video.cuda())
    if sdepth == 1 and tdepth == 0:
        show('patch tokens', tok_g.view_as(tok_b), tok_b, tok_f)
    enc_g = cv.encode(tok_b.cuda())
    show(f'encode depth {sdepth}+{tdepth} (teacher-forced tokens)', enc_g, enc_b, enc_f)
    del cv

_, mgc, crc = oracle_cfgs(FULL)
gen = torch.Generator().manual_seed(77)
ids = torch.randint(0, 65537, (1, 576), generator=gen)
ids[:, ::3] = 65536
ctx = weights.synthetic_context(1, 12, 768, seed=1, pad_last=3)
tm = (ctx != 0).any(-1)
for depth in (1, 2, 6):
    cfg = {**FULL['maskgit'], 'depth': depth}
    mg = P.MaskGit(**cfg)
    mg.load_state_dict(mg_sd, strict=False)
    mg = mg.cuda().eval()
    P.set_compute_dtype(mg, 'bf16')
    oc = {**mgc, 'depth': depth}
    kw = dict(video_patch_shape=(9, 8, 8), context=ctx, text_mask=tm)
    e_f = O.maskgit_forward(mg_sd, oc, ids, return_embeds=True, **kw)
    with O.precision('bf16'):
        e_b = O.maskgit_forward(mg_sd, oc, ids, return_embeds=True, **kw)
        l_b = O.maskgit_cfg(mg_sd, oc, ids, cond_scale=5., **kw)
    l_f = O.maskgit_cfg(mg_sd, oc, ids, cond_scale=5., **kw)
    e_g = mg(ids.cuda(), return_embeds=True, video_patch_shape=(9, 8, 8), context=ctx.cuda(), text_mask=tm.cuda())
    show(f'maskgit embeds depth {depth}', e_g, e_b, e_f)
    l_g = mg.forward_with_cond_scale(ids.cuda(), cond_scale=5., video_patch_shape=(9, 8, 8), context=ctx.cuda(), text_mask=tm.cuda())
    show(f'maskgit cfg logits depth {depth}', l_g, l_b, l_f)
    del mg
