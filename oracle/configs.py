"""Shared test configurations (TEST INFRASTRUCTURE ONLY).

TINY  -- seconds on a CPU; exercises every code path (odd FF inner dim 341, LFQ dim 8,
         2 heads, priming, text padding).
FULL  -- BASELINE.json's configuration (README.md:63-141 of the reference): dim 512,
         codebook 65 536, 256x256 / patch 32 / temporal patch 2, depth 4+4, MaskGit /
         TokenCritic depth 6, dim_context 768.
"""
import torch

from oracle import weights

TINY = dict(
    cvivit=dict(dim=128, codebook_size=256, image_size=64, patch_size=16, temporal_patch_size=2,
                spatial_depth=2, temporal_depth=2, dim_head=64, heads=2),
    maskgit=dict(dim=128, num_tokens=256, max_seq_len=128, depth=2, heads=2, dim_head=64, dim_context=96),
    critic=dict(dim=128, num_tokens=256, max_seq_len=128, depth=2, heads=2, dim_head=64, dim_context=96,
                has_cross_attn=True),
    steps=6,
)

FULL = dict(
    cvivit=dict(dim=512, codebook_size=65536, image_size=256, patch_size=32, temporal_patch_size=2,
                spatial_depth=4, temporal_depth=4, dim_head=64, heads=8),
    maskgit=dict(dim=512, num_tokens=65536, max_seq_len=1024, depth=6, heads=8, dim_head=64, dim_context=768),
    critic=dict(dim=512, num_tokens=65536, max_seq_len=1024, depth=6, heads=8, dim_head=64, dim_context=768,
                has_cross_attn=True),
    steps=18,
)


def oracle_cfgs(cfgs):
    """config dicts in the form oracle/phenaki_oracle.py expects."""
    c = cfgs['cvivit']
    cv = dict(image_size=(c['image_size'],) * 2, patch_size=(c['patch_size'],) * 2,
              temporal_patch_size=c['temporal_patch_size'], spatial_depth=c['spatial_depth'],
              temporal_depth=c['temporal_depth'], heads=c['heads'], channels=3)
    m = cfgs['maskgit']
    mg = dict(depth=m['depth'], heads=m['heads'], num_tokens=m['num_tokens'], unconditional=False)
    k = cfgs['critic']
    cr = dict(depth=k['depth'], heads=k['heads'], has_cross_attn=k['has_cross_attn'])
    return cv, mg, cr


def build_modules(ns, cfgs, with_phenaki=True, with_critic=True, steps=None):
    """construct CViViT / MaskGit / TokenCritic (/ Phenaki) from namespace ``ns`` -- the reference
    (oracle.ref_shim.load()) or the product package -- and fill them with the name-keyed weights."""
    cv = ns.CViViT(use_vgg_and_gan=False, **cfgs['cvivit'])
    mg = ns.MaskGit(**cfgs['maskgit'])
    cr = ns.TokenCritic(**cfgs['critic']) if with_critic else None
    weights.fill_module(cv, salt=1)
    weights.fill_module(mg, salt=2)
    if cr is not None:
        weights.fill_module(cr, salt=3)
    cv.eval(); mg.eval()
    ph = None
    if with_phenaki:
        ph = ns.Phenaki(maskgit=mg, cvivit=cv, critic=cr, steps=steps or cfgs['steps'],
                        text_embed_dim=cfgs['maskgit']['dim_context'])
        ph.eval()
    return cv, mg, cr, ph


build_reference = build_modules


def golden_dir():
    import os
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def state_dicts(tag):
    """(cvivit, maskgit, critic) CPU fp32 state_dicts with the name-keyed weights, built from the
    committed key/shape contract tests/golden/state_dict_keys.json (no module needed)."""
    import json
    import os
    with open(os.path.join(golden_dir(), 'state_dict_keys.json')) as f:
        keys = json.load(f)
    out = []
    for salt, kind in enumerate(('cvivit', 'maskgit', 'critic'), start=1):
        sd = {}
        for name, (shape, dtype) in keys[f'{tag}.{kind}'].items():
            dt = getattr(torch, dtype.split('.')[-1])
            ref = torch.empty(shape, dtype=dt)
            v = weights.fill_value(name, ref, salt)
            if v is None:
                assert name == 'vq.mask', name
                v = 2 ** torch.arange(shape[0] - 1, -1, -1)
            sd[name] = v.to(dt)
        out.append(sd)
    return out


def gan_state_dict(tag, discr_keys, requires_grad=False):
    """the C-ViViT state_dict of state_dicts(tag) plus the discriminator's entries (`discr.*`: name -> shape, as stored in
    tests/golden/gan_<tag>.pt) with the same name-keyed weights the golden generator gave the reference module (salt 1)"""
    sd = dict(state_dicts(tag)[0])
    for name, shape in discr_keys.items():
        sd[name] = weights.fill_value(name, torch.empty(shape), 1)
    if requires_grad:
        sd = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith('.beta') else v) for k, v in sd.items()}
    return sd
