"""sub-block diagnostic: product Attention / FeedForward / PEG blocks (bf16) vs oracle.precision('bf16') on random inputs"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_grad_enabled(False)
from oracle import phenaki_oracle as O, weights, hostcpu            # noqa: E402
from oracle.configs import FULL, oracle_cfgs, state_dicts          # noqa: E402
import phenaki_pytorch_amd as P                                      # noqa: E402
from phenaki_pytorch_amd import _lib as L                            # noqa: E402

hostcpu.configure()


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    d = (a - b)
    return d.abs().max().item() / b.abs().max().item(), (d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


def show(what, gpu, ob, of):
    (m1, r1), (m3, r3) = rel(gpu, ob), rel(ob, of)
    print(f'{what:40s} gpu-vs-oracle_bf16 max {m1:.2e} rms {r1:.2e} | oracle_bf16-vs-f32 max {m3:.2e} rms {r3:.2e}', flush=True)


cv_sd, mg_sd, cr_sd = state_dicts('full')
cfg = FULL
cv = P.CViViT(use_vgg_and_gan=False, **cfg['cvivit'])
cv.load_state_dict(cv_sd)
cv = cv.cuda().eval()
P.set_compute_dtype(cv, 'bf16')
g = torch.Generator().manual_seed(3)

# temporal self-attention (causal, ALiBi), S = 128, n = 9
S, n, D = 128, 9, 512
x = torch.randn(S, n, D, generator=g) * 1.5 + 0.1
att = cv.enc_temporal_transformer.layers[0][1]
p = 'enc_temporal_transformer.layers.0.1.'
of = O.attention(cv_sd, p, x, heads=8, causal=True)
with O.precision('bf16'):
    ob = O.attention(cv_sd, p, x, heads=8, causal=True)
gout = att.run(x.reshape(S * n, D).cuda(), S, n, L.BF16) - x.reshape(S * n, D).cuda()
show('temporal self-attn n=9 causal', gout.view(S, n, D), ob, of)

# spatial self-attention (bias), S = 18, n = 64
S, n = 18, 64
x = torch.randn(S, n, D, generator=g) * 1.5 + 0.1
att = cv.enc_spatial_transformer.layers[0][1]
p = 'enc_spatial_transformer.layers.0.1.'
bias = O.continuous_position_bias(cv_sd, 'spatial_rel_pos_bias.', (8, 8))
of = O.attention(cv_sd, p, x, heads=8, attn_bias=bias)
with O.precision('bf16'):
    ob = O.attention(cv_sd, p, x, heads=8, attn_bias=bias)
gout = att.run(x.reshape(S * n, D).cuda(), S, n, L.BF16, attn_bias=cv.spatial_rel_pos_bias(8, 8)) - x.reshape(S * n, D).cuda()
show('spatial self-attn n=64 bias', gout.view(S, n, D), ob, of)

# feed-forward
ff = cv.enc_spatial_transformer.layers[0][3]
p = 'enc_spatial_transformer.layers.0.3.'
x2 = torch.randn(1152, D, generator=g) * 1.5 + 0.1
of = O.feedforward(cv_sd, p, x2)
with O.precision('bf16'):
    ob = O.feedforward(cv_sd, p, x2)
gout = ff.run(x2.cuda(), L.BF16) - x2.cuda()
show('feed-forward', gout, ob, of)

# PEG (temporal, causal) on the scrambled view
peg = cv.enc_temporal_transformer.layers[0][0]
p = 'enc_temporal_transformer.layers.0.0.'
xs = torch.randn(128, 9, D, generator=g)
of = O.peg(cv_sd, p, xs, (2, 9, 8, 8), True)
gout = peg.run(xs.reshape(-1, D).cuda(), (2, 9, 8, 8)) - xs.reshape(-1, D).cuda()
show('PEG causal', gout.view(128, 9, D), of, of)

# maskgit self-attn n = 576 with CPB bias, and cross-attn on a 12-token context
mg = P.MaskGit(**cfg['maskgit'])
mg.load_state_dict(mg_sd)
mg = mg.cuda().eval()
P.set_compute_dtype(mg, 'bf16')
S, n = 2, 576
x = torch.randn(S, n, D, generator=g) * 1.5 + 0.1
att = mg.transformer.layers[0][1]
p = 'transformer.layers.0.1.'
bias = O.continuous_position_bias(mg_sd, 'continuous_pos_bias.', (9, 8, 8))
of = O.attention(mg_sd, p, x, heads=8, attn_bias=bias)
with O.precision('bf16'):
    ob = O.attention(mg_sd, p, x, heads=8, attn_bias=bias)
gout = att.run(x.reshape(S * n, D).cuda(), S, n, L.BF16, attn_bias=mg.continuous_pos_bias(9, 8, 8)) - x.reshape(S * n, D).cuda()
show('maskgit self-attn n=576 bias', gout.view(S, n, D), ob, of)

ctx = weights.synthetic_context(2, 12, 768, seed=1, pad_last=3)
tm = (ctx != 0).any(-1)
att = mg.transformer.layers[0][2]
p = 'transformer.layers.0.2.'
of = O.attention(mg_sd, p, x, heads=8, context=ctx, mask=tm)
with O.precision('bf16'):
    ob = O.attention(mg_sd, p, x, heads=8, context=ctx, mask=tm)
for cached in (False, True):
    cache = {}
    for rep in range(2 if cached else 1):
        gout = att.run(x.reshape(S * n, D).cuda(), S, n, L.BF16, context2d=ctx.reshape(-1, 768).cuda(), n_ctx=12, kmask=tm.to(torch.uint8).cuda(),
                       kv_cache=cache) - x.reshape(S * n, D).cuda()
    show(f'maskgit cross-attn (kv cached: {cached})', gout.view(S, n, D), ob, of)
