"""GPU parity tests of the training-step kernels: every autograd block of phenaki_pytorch_amd/train.py against torch
autograd of the oracle's restatement of the same reference lines on CPU, f32 / split-bf16 at 1e-3 (bf16: its own tolerance); the whole
`Phenaki.forward` gradient against the REAL reference's autograd is in test_modules_gpu.py (golden)."""
import pytest
import torch
import torch.nn.functional as F

import os

from oracle import phenaki_oracle as O
from oracle import weights
from oracle.configs import FULL, TINY, oracle_cfgs, state_dicts
from tests.util import close, load_product, record_parity

pytestmark = pytest.mark.gpu

MODES = [('fp32', 1e-3), ('bf16x3', 1e-3), ('bf16', 6e-2)]


@pytest.fixture(autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    torch.cuda.set_device(0)
    with torch.enable_grad():                      # other test modules switch grad mode off process-wide at import
        yield


def _leaf(t):
    return t.clone().requires_grad_()


def test_pack_colsum_scatter_kernels():
    from phenaki_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(0)
    src = torch.randn(133, 71, generator=g).cuda()
    for tr in (False, True):
        R, K = (71, 133) if tr else (133, 71)
        want = (src.t() if tr else src).contiguous()
        for kind, q in ((0, 32), (1, 64), (2, 32)):
            Kp = (K + q - 1) // q * q
            out = torch.full((R, Kp), 7.0, device='cuda', dtype=torch.bfloat16 if kind == 1 else torch.float32)
            L.pack(src, R, K, tr, out, Kp, kind)
            if kind == 2:                                        # (hi | lo) planes per 32-element block
                img = out.view(torch.bfloat16).view(R, Kp // 32, 2, 32).float()
                got = (img[:, :, 0] + img[:, :, 1]).reshape(R, Kp)
                tol = 2e-5
            else:
                got, tol = out.float(), (0 if kind == 0 else 4e-3)
            assert (got[:, K:] == 0).all(), 'pad columns must be zero'
            assert (got[:, :K] - want).abs().max() <= tol * want.abs().max()
    rows = torch.tensor([5, 0, 132, 7, 7], dtype=torch.int32).cuda()
    out = torch.empty((5, 96), device='cuda')
    L.pack(src, 5, 71, False, out, 96, 0, rows=rows)
    assert torch.equal(out[:, :71], src[rows.long()]) and (out[:, 71:] == 0).all()
    out = torch.empty((71, 32), device='cuda')
    L.pack(src, 71, 5, True, out, 32, 0, rows=rows)
    assert torch.equal(out[:, :5], src[rows.long()].t())
    big = torch.randn(5000, 200, generator=g).cuda()
    cs = L.colsum(big, 5000, 200, torch.empty(200, device='cuda'), scale=0.5)
    close(cs.cpu(), 0.5 * big.cpu().double().sum(0).float(), 1e-5, 'colsum')
    cs2 = L.colsum(big, 5000, 200, cs.clone(), accumulate=True)
    close(cs2.cpu(), 1.5 * big.cpu().double().sum(0).float(), 1e-5, 'colsum accumulate')
    dst = torch.zeros((133, 72), device='cuda')
    r4 = torch.tensor([3, 100, 9], dtype=torch.int32).cuda()
    s4 = torch.randn(3, 72, generator=g).cuda()
    L.scatter_rows(s4, r4, dst, 3, 72)
    assert torch.equal(dst[r4.long()], s4) and float(dst.abs().sum()) == pytest.approx(float(s4.abs().sum()), rel=1e-6)


def _unsplit(out, kind, R, Kp):
    """an operand image of pk_pack back as f32 (kind 2: hi + lo planes of each 32-element block)"""
    if kind == 2:
        img = out.view(torch.bfloat16).view(R, Kp // 32, 2, 32).float()
        return (img[:, :, 0] + img[:, :, 1]).reshape(R, Kp)
    return out.float()


def test_multi_job_pack_sum_and_one_launch_colsum():
    """round 6: pk_pack_multi / pk_pack_table (several operand images per launch, incl. column / row blocks inside a larger image),
    pk_sum_batch_multi, and pk_colsum's one-launch form -- each against the single-job entry point or a float64 sum"""
    from phenaki_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    mats = [torch.randn(r, c, generator=g).cuda() for r, c in ((133, 71), (64, 64), (300, 1365), (7, 520), (1365, 96), (96, 40), (200, 33), (65, 129), (31, 17), (512, 512))]
    for kind, q in ((0, 32), (1, 64), (2, 32)):
        td = torch.bfloat16 if kind == 1 else torch.float32
        jobs, outs, want = [], [], []
        for i, m in enumerate(mats):
            tr = i % 2 == 1
            R, K = (m.shape[1], m.shape[0]) if tr else tuple(m.shape)
            Kp = (K + q - 1) // q * q
            out = torch.full((R, Kp), 7.0, device='cuda', dtype=td)
            ref = torch.full((R, Kp), 7.0, device='cuda', dtype=td)
            L.pack(m, R, K, tr, ref, Kp, kind)
            jobs.append(L.pack_job(m, R, K, tr, out, Kp, kind))
            outs.append(out)
            want.append(ref)
        L.pack_multi(jobs, mats[0])                                 # 10 jobs: two launches (8 + 2)
        for o, w in zip(outs, want):
            assert torch.equal(o.view(torch.int16 if kind == 1 else torch.int32), w.view(torch.int16 if kind == 1 else torch.int32))
        # the device-table form, with blocks inside a zero-initialised image: rows [0, 40) and [48, 88) <- two matrices, and a transposed column block
        a, b = torch.randn(40, 100, generator=g).cuda(), torch.randn(40, 100, generator=g).cuda()
        Kp = (100 + q - 1) // q * q
        img = torch.zeros((96, Kp), device='cuda', dtype=td)
        Kq = (96 + q - 1) // q * q
        imgT = torch.zeros((100, Kq), device='cuda', dtype=td)
        tab = L.PackTable([L.pack_job(a, 40, 100, False, img[:40], Kp, kind), L.pack_job(b, 40, 100, False, img[48:88], Kp, kind),
                           L.pack_job(a, 100, 40, True, imgT[:, :48], 40, kind), L.pack_job(b, 100, 40, True, imgT[:, 48:96], 40, kind)], 'cuda')
        for _ in range(2):                                           # replayed: same bytes
            tab.run()
        tol = 0 if kind == 0 else (4e-3 if kind == 1 else 2e-5)
        got, gotT = _unsplit(img, kind, 96, Kp), _unsplit(imgT, kind, 100, Kq)
        full = torch.zeros((96, 100), device='cuda')
        full[:40], full[48:88] = a, b
        assert (got[:, :100] - full).abs().max() <= tol * 4 and (got[:, 100:] == 0).all()
        assert (gotT[:, :96] - full.t()).abs().max() <= tol * 4 and (gotT[:, 96:] == 0).all()
    parts = [torch.randn(s, e, generator=g).cuda() for s, e in ((8, 512 * 512), (4, 1368 * 512), (2, 64), (8, 1365 * 512))]
    outs = [torch.empty(p_.shape[1], device='cuda') for p_ in parts]
    L.sum_batch_multi([(p_, p_.shape[0], o, p_.shape[1]) for p_, o in zip(parts, outs)])
    for p_, o in zip(parts, outs):
        ref = torch.empty_like(o)
        L.sum_batch(p_, p_.shape[0], ref, p_.shape[1])
        assert torch.equal(o, ref)
    srcs = [torch.randn(m_, n_, generator=g).cuda() for m_, n_ in ((576, 512), (1024, 128), (288, 27 * 64), (4608, 512), (96, 768))]
    outs = [torch.empty(x_.shape[1], device='cuda') for x_ in srcs]
    L.colsum_multi([(x_, x_.shape[0], x_.shape[1], o) for x_, o in zip(srcs, outs)])
    for x_, o in zip(srcs, outs):
        close(o.cpu(), x_.cpu().double().sum(0).float(), 1e-5, f'colsum_multi {tuple(x_.shape)}')
        ref = L.colsum(x_, x_.shape[0], x_.shape[1], torch.empty(x_.shape[1], device='cuda'))
        assert torch.equal(o, ref), 'the multi-job form adds in the same order as the one-launch form'
    for M, N in ((144, 512), (288, 27 * 512), (1024, 128), (144, 1), (200, 33), (1000, 1024), (9000, 64), (9000, 6)):
        src = torch.randn(M, N, generator=g).cuda()
        cs = L.colsum(src, M, N, torch.empty(N, device='cuda'), scale=0.25)
        close(cs.cpu(), 0.25 * src.cpu().double().sum(0).float(), 1e-5, f'colsum one launch {M}x{N}')
        cs2 = L.colsum(src, M, N, cs.clone(), accumulate=True)
        close((cs2 - cs).cpu(), src.cpu().double().sum(0).float(), 1e-4, f'colsum one launch accumulate {M}x{N}')


@pytest.mark.parametrize('D', [512, 768, 64])
def test_layernorm_backward(D):
    from phenaki_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(1)
    M = 333
    x, gam, bet = torch.randn(M, D, generator=g) * 2 + 0.5, torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
    dy, add = torch.randn(M, D, generator=g), torch.randn(M, D, generator=g)
    xl, gl, bl = _leaf(x), _leaf(gam), _leaf(bet)
    F.layer_norm(xl, (D,), gl, bl).backward(dy)
    dx = torch.empty((M, D), device='cuda')
    dg, db = L.layernorm_bwd(x.cuda(), gam.cuda(), dy.cuda(), dx, M, D, add=add.cuda(), want_beta=True)
    close(dx.cpu(), xl.grad + add, 1e-4, 'ln dx')
    close(dg.cpu(), gl.grad, 1e-4, 'ln dgamma')
    close(db.cpu(), bl.grad, 1e-4, 'ln dbeta')


@pytest.mark.parametrize('dtype', ['fp32', 'bf16x3'])
def test_weight_images_follow_the_parameters(dtype):
    """round 6: the persistent W / W^T operand images of a Transformer (train.WeightImages) are refreshed every pass and rebuilt when a parameter
    moves to other storage; deepcopy / state_dict never see them.  A stale image would reproduce the OLD gradients."""
    import copy
    import phenaki_pytorch_amd as P
    from phenaki_pytorch_amd import train as T
    from phenaki_pytorch_amd.attention import resolve_dtype
    torch.manual_seed(5)
    D, S, n = 128, 3, 40
    tr = P.attention.Transformer(dim=D, depth=2, heads=2, has_cross_attn=True, dim_context=96).cuda()
    dt = resolve_dtype(dtype)
    x = torch.randn(S * n, D).cuda()
    ctx = torch.randn(S * 5, 96).cuda()
    G = torch.randn(S * n, D).cuda()

    def grads(mod):
        for p_ in mod.parameters():
            p_.grad = None
        xc = x.clone().requires_grad_()
        with torch.enable_grad():
            y = T.transformer_train(mod, xc, S, n, dt, context2d=ctx, n_ctx=5)
        y.backward(G)
        return {k: v.grad.clone() for k, v in mod.named_parameters() if v.grad is not None}, y.detach()

    g1, y1 = grads(tr)
    g1b, y1b = grads(tr)                                               # replay on the same images: same bytes
    assert torch.equal(y1, y1b) and all(torch.equal(g1[k], g1b[k]) for k in g1)
    assert not any('image' in k or 'pk_' in k for k in tr.state_dict())
    # in-place update (what the optimizer does): same storage, new values -> the refresh must pick them up
    with torch.no_grad():
        for p_ in tr.parameters():
            if p_.ndim == 2:
                p_.mul_(1.25)
    fresh = copy.deepcopy(tr)                                          # a new module: its own images, built from the updated values
    g2, y2 = grads(tr)
    gf, yf = grads(fresh)
    assert not torch.equal(y1, y2)
    assert torch.equal(y2, yf) and all(torch.equal(g2[k], gf[k]) for k in g2), 'images not refreshed after an in-place parameter update'
    # new storage (what .to() / a reload by assignment does) with new values
    with torch.no_grad():
        for p_ in tr.parameters():
            if p_.ndim == 2:
                p_.data = (p_.data * 0.8).clone()
    fresh = copy.deepcopy(tr)
    g3, y3 = grads(tr)
    gf, yf = grads(fresh)
    assert torch.equal(y3, yf) and all(torch.equal(g3[k], gf[k]) for k in g3), 'images not rebuilt after the parameters moved to other storage'


@pytest.mark.parametrize('dtype,tol', MODES)
def test_feedforward_block_gradients(dtype, tol):
    """x + FeedForward(x) (attention.py:45-52, inner 1365 -> padded 1368 inside the block) vs torch autograd of the oracle"""
    import phenaki_pytorch_amd as P
    from phenaki_pytorch_amd.train import feedforward_train
    from phenaki_pytorch_amd.attention import resolve_dtype
    torch.manual_seed(2)
    D, M = 512, 200
    ff = P.attention.FeedForward(dim=D)
    with torch.no_grad():
        ff[0].weight.uniform_(0.5, 1.5)
        ff[0].bias.normal_(0, 0.3)
    sd = {k: _leaf(v.detach()) for k, v in ff.state_dict().items()}
    x, G = torch.randn(M, D), torch.randn(M, D)
    xl = _leaf(x)
    (O.feedforward(sd, '', xl[None])[0] + xl).backward(G)
    ff = ff.cuda()
    xc = x.cuda().requires_grad_()
    with torch.enable_grad():
        y = feedforward_train(ff, xc, resolve_dtype(dtype))
    y.backward(G.cuda())
    errs = dict(dx=close(xc.grad.cpu(), xl.grad, tol, 'ff dx'),
                dw1=close(ff[1].weight.grad.cpu(), sd['1.weight'].grad, tol, 'ff dW1'),
                dw2=close(ff[4].weight.grad.cpu(), sd['4.weight'].grad, tol, 'ff dW2'),
                dlnw=close(ff[0].weight.grad.cpu(), sd['0.weight'].grad, tol, 'ff d ln weight'),
                dlnb=close(ff[0].bias.grad.cpu(), sd['0.bias'].grad, tol, 'ff d ln bias'))
    record_parity('train_ff_block', dict(dtype=dtype, **errs))


@pytest.mark.parametrize('dtype,tol', MODES)
@pytest.mark.parametrize('case', ['self_bias', 'self_bias_long', 'self_long', 'cross_null_mask', 'self_mask', 'causal_short', 'causal_long', 'causal_packed', 'packed_mask'])
def test_attention_block_gradients(case, dtype, tol):
    """x + Attention(x) (attention.py:89-182): position bias gradient, null keys, key mask, l2norm / learned scales; n = 70 (ragged tiles);
    causal_*: ALiBi over the null + real keys and the causal mask (the C-ViViT temporal transformers: n = 9, and n = 70 across key tiles);
    *packed*: no null keys and n <= 32 -- pk_attn_bwd packs 64 / n whole (sequence, head) groups into one tile (80 groups of 9 rows = 11 full
    tiles + a partial one; 22 groups of 12 rows with a key mask); *_long (round 6): n = 200 >= 128 without a key mask -- the training forward runs on the
    LDS-staged kernels there (split-bf16: running-max form with the bias as a matrix, ragged last key tile) and hands the backward its log-sum-exp"""
    import phenaki_pytorch_amd as P
    from phenaki_pytorch_amd.train import attention_train
    from phenaki_pytorch_amd.attention import resolve_dtype
    torch.manual_seed(3)
    D, heads, S, n = 128, 2, 3, 70
    cross = case == 'cross_null_mask'
    causal = case.startswith('causal')
    packed = 'packed' in case
    if case in ('causal_short', 'causal_packed'):
        S, n = 40, 9
    elif case == 'packed_mask':
        S, n = 11, 12
    elif case.endswith('_long'):
        S, n = 2, 200
    nnull = 2 if (cross or (causal and not packed)) else 0
    attn = P.attention.Attention(dim=D, dim_context=96 if cross else None, heads=heads, num_null_kv=nnull, causal=causal)
    with torch.no_grad():
        attn.q_scale.uniform_(0.5, 1.5)
        attn.k_scale.uniform_(0.5, 1.5)
        attn.norm.gamma.uniform_(0.5, 1.5)
        if cross:
            attn.context_norm.gamma.uniform_(0.5, 1.5)
    sd = {k: _leaf(v.detach()) for k, v in attn.state_dict().items()}
    sd['norm.beta'] = attn.norm.beta.detach()
    x, G = torch.randn(S, n, D), torch.randn(S, n, D)
    xl = _leaf(x)
    ctx = mask = bias = None
    n_ctx = None
    if cross:
        n_ctx = 13
        ctx = torch.randn(S, n_ctx, 96)
        mask = torch.rand(S, n_ctx) > 0.3
        mask[:, 0] = True
        sd['context_norm.beta'] = attn.context_norm.beta.detach()
    elif case in ('self_bias', 'self_bias_long'):
        bias = _leaf(torch.randn(heads, n, n))
    elif case == 'self_long':
        pass
    elif not causal:
        mask = torch.rand(S, n) > 0.2
        mask[:, 0] = True
    (O.attention(sd, '', xl, heads=heads, context=ctx, mask=mask, attn_bias=bias, causal=causal) + xl).backward(G)
    attn = attn.cuda()
    xc = x.reshape(S * n, D).cuda().requires_grad_()
    bc = bias.detach().cuda().requires_grad_() if bias is not None else None
    with torch.enable_grad():
        y = attention_train(attn, xc, S, n, resolve_dtype(dtype), context2d=ctx.reshape(S * n_ctx, 96).cuda() if cross else None, n_ctx=n_ctx,
                            attn_bias=bc, kmask=mask.to(torch.uint8).cuda() if mask is not None else None)
    y.backward(G.reshape(S * n, D).cuda())
    errs = dict(dx=close(xc.grad.cpu(), xl.grad.reshape(S * n, D), tol, 'attn dx'))
    for name in ('to_q.weight', 'to_kv.weight', 'to_out.weight', 'q_scale', 'k_scale', 'norm.gamma') + (('null_kv',) if nnull else ()) + \
            (('context_norm.gamma',) if cross else ()):
        mod = attn
        for part in name.split('.'):
            mod = getattr(mod, part)
        errs[name] = close(mod.grad.cpu(), sd[name].grad, tol, f'attn d {name}')
    if bias is not None:
        errs['bias'] = close(bc.grad.cpu(), bias.grad, tol, 'attn d bias')
    record_parity('train_attention_block', dict(case=case, dtype=dtype, **errs))


@pytest.mark.parametrize('causal', [False, True])
def test_peg_block_gradients(causal):
    import phenaki_pytorch_amd as P
    from phenaki_pytorch_amd.train import peg_train
    torch.manual_seed(4)
    D, shape = 64, (2, 5, 4, 3)
    peg = P.attention.PEG(dim=D, causal=causal)
    sd = {k: _leaf(v.detach()) for k, v in peg.state_dict().items()}
    M = shape[0] * shape[1] * shape[2] * shape[3]
    x, G = torch.randn(M, D), torch.randn(M, D)
    xl = _leaf(x)
    (O.peg(sd, '', xl.reshape(shape[0], -1, D), shape, causal).reshape(M, D) + xl).backward(G)
    peg = peg.cuda()
    xc = x.cuda().requires_grad_()
    with torch.enable_grad():
        y = peg_train(peg, xc, shape)
    y.backward(G.cuda())
    close(xc.grad.cpu(), xl.grad, 1e-4, 'peg dx')
    close(peg.dsconv.weight.grad.cpu(), sd['dsconv.weight'].grad, 1e-4, 'peg d weight')
    close(peg.dsconv.bias.grad.cpu(), sd['dsconv.bias'].grad, 1e-4, 'peg d bias')


@pytest.mark.parametrize('heads', [4, 2, 8])
def test_position_bias_gradients(heads):
    """ContinuousPositionBias on the relative-position table (prod(2 d - 1) rows instead of n^2) == the reference's n^2 form, values and gradients"""
    import phenaki_pytorch_amd as P
    from phenaki_pytorch_amd.train import position_bias_train
    torch.manual_seed(5)
    dims = (3, 4, 2)
    cpb = P.attention.ContinuousPositionBias(dim=32, heads=heads, num_dims=3)
    sd = {k: _leaf(v.detach()) for k, v in cpb.state_dict().items()}
    n = 24
    G = torch.randn(heads, n, n)
    ref = O.continuous_position_bias(sd, '', dims)
    ref.backward(G)
    cpb = cpb.cuda()
    with torch.enable_grad():
        out = position_bias_train(cpb, dims, torch.device('cuda'))
    close(out.detach().cpu(), ref.detach(), 1e-5, 'position bias value')
    out.backward(G.cuda())
    for k, v in cpb.named_parameters():
        close(v.grad.cpu(), sd[k].grad, 2e-4, f'position bias d {k}')


def test_embed_and_bce_head_gradients():
    from phenaki_pytorch_amd.train import _BCEHead, _Embed
    g = torch.Generator().manual_seed(6)
    V1, P_, D, b, n = 50, 40, 64, 3, 17
    tok, pos = torch.randn(V1, D, generator=g), torch.randn(P_, D, generator=g)
    ids = torch.randint(0, V1, (b, n), generator=g)
    ids[:, ::3] = V1 - 1
    G = torch.randn(b * n, D, generator=g)
    tl, pl = _leaf(tok), _leaf(pos)
    xr = (tl[ids] + pl[:n][None]).reshape(b * n, D)
    (xr * 0.1 + xr.detach() * 0.9).backward(G)
    tc, pc = tok.cuda().requires_grad_(), pos.cuda().requires_grad_()
    with torch.enable_grad():
        x = _Embed.apply(tc, pc, ids.cuda(), 0.1)
    close(x.detach().cpu(), xr.detach(), 1e-6, 'embed value')
    x.backward(G.cuda())
    close(tc.grad.cpu(), tl.grad, 1e-5, 'd token_emb')
    close(pc.grad.cpu(), pl.grad, 1e-5, 'd pos_emb')
    # critic head + BCE
    M = 77
    e, w, bb = torch.randn(M, D, generator=g), torch.randn(1, D, generator=g) * 0.3, torch.randn(1, generator=g)
    y = (torch.rand(M, generator=g) > 0.5).float()
    el, wl, bl = _leaf(e), _leaf(w), _leaf(bb)
    ref = F.binary_cross_entropy_with_logits((el @ wl.t()).squeeze(-1) + bl, y)
    (ref * 1.7).backward()
    ec, wc, bc = e.cuda().requires_grad_(), w.cuda().requires_grad_(), bb.cuda().requires_grad_()
    with torch.enable_grad():
        loss = _BCEHead.apply(ec, wc, bc, y.cuda())
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach()))
    (loss * 1.7).backward()
    close(ec.grad.cpu(), el.grad, 1e-4, 'bce de')
    close(wc.grad.cpu(), wl.grad, 1e-4, 'bce dw')
    close(bc.grad.cpu(), bl.grad, 1e-4, 'bce db')


@pytest.mark.parametrize('dtype,tol', [('fp32', 1e-3), ('bf16x3', 1e-3), ('bf16', 1e-1)])
def test_training_step_matches_reference_autograd(golden_dir, dtype, tol):
    """loss = Phenaki.forward(...); loss.backward() against the REAL reference's autograd (tiny config, the reference's
    three random draws injected): the loss and the gradient of all 100 MaskGit + TokenCritic parameters; generator-only / critic-only
    variants train exactly the parameters the reference trains."""
    g = torch.load(os.path.join(golden_dir, 'forward_grads_tiny.pt'), weights_only=False)
    cv, mg, cr, ph = load_product('tiny', TINY, dtype=dtype)
    batch = g['batch']
    ctx = weights.synthetic_context(batch, g['ctx_len'], TINY['maskgit']['dim_context'], seed=3, pad_last=2).cuda()
    ids = g['ids'].cuda()
    n = ids[0].numel()
    draws = dict(rand_step=g['rand_step'], perm_noise=weights.uniform_noise((batch, n), 700),
                 gumbel_u=weights.uniform_noise((batch, n, TINY['maskgit']['num_tokens']), 701))
    named = {f'maskgit.{k}': v for k, v in mg.named_parameters()}
    named.update({f'critic.{k}': v for k, v in cr.named_parameters()})

    def step(**kw):
        for v in named.values():
            v.grad = None
        loss = ph(video_codebook_ids=ids, text_embeds=ctx, _draws=draws, **kw)
        loss.backward()
        return loss.detach()

    loss = step()
    assert abs(float(loss) - float(g['loss_total'])) <= min(tol, 2e-2) * abs(float(g['loss_total']))
    assert sorted(k for k, v in named.items() if v.grad is not None) == sorted(g['grads_total'])
    errs = {}
    top = max(float(v.abs().max()) for v in g['grads_total'].values() if v.numel())
    for k, ref in g['grads_total'].items():
        if ref.numel() == 0:
            continue
        if dtype == 'bf16' and k.startswith('critic.'):
            # bf16 logits flip a few gumbel-argmax near-ties (audited in the sampler tests): the critic then sees other input ids than the
            # reference's critic did, so its gradients are those of a different batch -- finite is all that can be asked here
            assert torch.isfinite(named[k].grad).all()
            continue
        if float(ref.abs().max()) < 1e-6 * top:
            # structurally zero (the last position-bias Linear's bias shifts every score of a row by the same amount: softmax does not see
            # it) -- the reference's own value is rounding noise; ours must be noise of the same kind, measured against the real gradients
            assert float(named[k].grad.abs().max()) <= 1e-2 * tol * top, f'd {k} should vanish'
            continue
        errs[k] = close(named[k].grad.cpu(), ref, tol, f'd {k} ({dtype})')
    worst = max(errs, key=errs.get)
    record_parity('training_step_vs_reference_autograd', dict(dtype=dtype, parameters=len(errs), worst=worst, worst_rel_err=errs[worst],
                                                              median_rel_err=sorted(errs.values())[len(errs) // 2]))
    for name in ('generator', 'critic'):
        l2 = step(**{f'only_train_{name}': True})
        assert abs(float(l2) - float(g[f'loss_{name}'])) <= min(tol, 2e-2) * abs(float(g[f'loss_{name}']))
        got = sorted(k for k, v in named.items() if v.grad is not None)
        assert got == g[f'grad_keys_{name}'], f'only_train_{name}: parameters with a gradient differ from the reference'
        pk, pv = g[f'grad_probe_{name}']
        if not (dtype == 'bf16' and name == 'critic'):
            close(named[pk].grad.cpu(), pv, tol, f'only_train_{name}: d {pk}')


def test_hip_adamw_matches_torch_adamw_and_bumps_versions():
    """optim.py (reference optimizer.py:11-37): get_optimizer's grouping and the pk_adamw update against torch.optim.AdamW / Adam on CPU"""
    import phenaki_pytorch_amd as P
    g = torch.Generator().manual_seed(8)
    # pk_adamw_multi packs 40 tensors / 512 chunks of 2048 elements per launch and gives tensors of >= 256 Ki elements their own launch:
    # 45 small vectors (two launches), six 98-chunk tensors (a tensor split across launches), one large matrix, odd sizes
    shapes = [(33, 17), (129,), (4, 3, 3), (100, 90), (5000,), (7, 3), (1,)] + [(64,)] * 45 + [(200000,)] * 6 + [(600, 500)]
    for wd in (1e-2, 0.0):
        ref_p = [torch.randn(*s, generator=g).requires_grad_() for s in shapes]
        hip_p = [p.detach().clone().cuda().requires_grad_() for p in ref_p]
        if wd == 0:
            ref = torch.optim.Adam(ref_p, lr=3e-3, betas=(0.9, 0.99), eps=1e-8)
        else:
            wdp, nowd = [p for p in ref_p if p.ndim >= 2], [p for p in ref_p if p.ndim < 2]
            ref = torch.optim.AdamW([{'params': wdp}, {'params': nowd, 'weight_decay': 0}], lr=3e-3, weight_decay=wd, betas=(0.9, 0.99), eps=1e-8)
        opt = P.get_optimizer(hip_p, lr=3e-3, wd=wd)
        assert isinstance(opt, P.HipAdamW) and len(opt.param_groups) == (1 if wd == 0 else 2)
        for it in range(4):
            grads = [torch.randn(*s, generator=g) for s in shapes]
            for p, q, gr in zip(ref_p, hip_p, grads):
                p.grad, q.grad = gr.clone(), gr.cuda()
            v0 = [q._version for q in hip_p]
            ref.step()
            opt.step()
            assert all(q._version > v for q, v in zip(hip_p, v0)), 'the update must be visible to version-keyed caches'
            for p, q in zip(ref_p, hip_p):
                close(q.detach().cpu(), p.detach(), 1e-5, f'adamw step {it} wd {wd}')
        sd = opt.state_dict()
        assert len(sd['state']) == len(shapes) and sd['state'][0]['step'] == 4


@pytest.mark.parametrize('variant', ['self_critic', 'unconditional', 'no_critic_video_mask'])
def test_training_step_variants_run_and_match_oracle_autograd(variant):
    """the other constructions Phenaki.forward trains (phenaki_pytorch.py:353-372 self_token_critic, unconditional MaskGit, a video frame mask):
    loss and a few gradients against torch autograd through the oracle on the same draws"""
    import phenaki_pytorch_amd as P
    from oracle.configs import oracle_cfgs, state_dicts
    cv, mg, cr, ph = load_product('tiny', TINY)
    _, mg_sd, cr_sd = state_dicts('tiny')
    _, mgc, crc = oracle_cfgs(TINY)
    batch, n = 2, 48
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(0, TINY['maskgit']['num_tokens'], (batch, 3, 4, 4), generator=g)
    ctx = weights.synthetic_context(batch, 6, TINY['maskgit']['dim_context'], seed=3, pad_last=2)
    draws = dict(rand_step=torch.tensor([1, 3]), perm_noise=weights.uniform_noise((batch, n), 710),
                 gumbel_u=weights.uniform_noise((batch, n, TINY['maskgit']['num_tokens']), 711))
    okw = dict(patch_shape=(3, 4, 4), steps=TINY['steps'], mask_id=TINY['maskgit']['num_tokens'], **draws)
    leaf = lambda sd: {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith('.beta') else v) for k, v in sd.items()}
    mgl = leaf(mg_sd)
    if variant == 'self_critic':
        ph2 = P.Phenaki(maskgit=mg, cvivit=cv, self_token_critic=True, steps=TINY['steps'], text_embed_dim=TINY['maskgit']['dim_context']).cuda()
        torch.manual_seed(0)
        with torch.no_grad():
            ph2.critic.to_pred[0].weight.normal_(0, 0.1)
        head = {'to_pred.0.weight': ph2.critic.to_pred[0].weight.detach().cpu().clone().requires_grad_(),
                'to_pred.0.bias': ph2.critic.to_pred[0].bias.detach().cpu().clone().requires_grad_()}
        ref = O.phenaki_forward_loss(mgl, mgc, head, dict(self_critic=(mgl, mgc)), ids.flatten(1), context=ctx, **okw)
        loss = ph2(video_codebook_ids=ids.cuda(), text_embeds=ctx.cuda(), _draws=draws)
        probes = [(mg.to_logits.weight, mgl['to_logits.weight']), (mg.token_emb.weight, mgl['token_emb.weight']),
                  (ph2.critic.to_pred[0].weight, head['to_pred.0.weight']), (mg.transformer.layers[0][2].null_kv, mgl['transformer.layers.0.2.null_kv'])]
    elif variant == 'unconditional':
        mgu = P.MaskGit(unconditional=True, **TINY['maskgit'])
        sdu = {k: v for k, v in mg_sd.items() if k in mgu.state_dict()}
        mgu.load_state_dict(sdu)
        mgu = mgu.cuda()
        ph2 = P.Phenaki(maskgit=mgu, cvivit=cv, steps=TINY['steps'], text_embed_dim=TINY['maskgit']['dim_context']).cuda()
        mgl = leaf(sdu)
        ref = O.phenaki_forward_loss(mgl, dict(mgc, unconditional=True), None, None, ids.flatten(1), context=None, **okw)
        loss = ph2(video_codebook_ids=ids.cuda(), _draws=draws)
        probes = [(mgu.to_logits.bias, mgl['to_logits.bias']), (mgu.pos_emb.weight, mgl['pos_emb.weight']),
                  (mgu.transformer.layers[1][3][1].weight, mgl['transformer.layers.1.3.1.weight'])]
    else:
        ph2 = P.Phenaki(maskgit=mg, cvivit=cv, steps=TINY['steps'], text_embed_dim=TINY['maskgit']['dim_context']).cuda()
        H = TINY['cvivit']['image_size']
        video = weights.synthetic_video(batch, 5, H, H, seed=5)
        fmask = torch.tensor([[True] * 5, [True] * 3 + [False] * 2])
        with torch.no_grad():
            vids = cv(video.cuda(), return_only_codebook_ids=True)
            vmask = cv.calculate_video_token_mask(video.cuda(), video_frame_mask=fmask.cuda()).cpu()
        ref = O.phenaki_forward_loss(mgl, mgc, None, None, vids.cpu().flatten(1), context=ctx, video_mask=vmask, **okw)
        loss = ph2(video.cuda(), text_embeds=ctx.cuda(), video_frame_mask=fmask.cuda(), _draws=draws)
        probes = [(mg.to_logits.weight, mgl['to_logits.weight']), (mg.transformer.layers[0][1].to_kv.weight, mgl['transformer.layers.0.1.to_kv.weight']),
                  (mg.continuous_pos_bias.net[0][0].weight, mgl['continuous_pos_bias.net.0.0.weight'])]
    ref['loss'].backward()
    loss.backward()
    assert abs(float(loss.detach()) - float(ref['loss'].detach())) <= 2e-4 * abs(float(ref['loss'].detach()))
    for got, want in probes:
        close(got.grad.cpu(), want.grad, 1e-3, f'{variant} gradient probe')


@pytest.mark.parametrize('dtype,tol', [('fp32', 1e-3), ('bf16x3', 1e-3)])
def test_training_step_full_config_matches_oracle_autograd(dtype, tol):
    """BASELINE geometry (dim 512, depth 6 + 6, 8 heads, n = 576 = 9 key tiles, vocab 65 536 in 32 slabs, inner 1365, context 12 + 2 null keys):
    loss.backward() of Phenaki.forward against torch autograd through the (reference-pinned) oracle on CPU, same three draws; all
    gradients of both networks"""
    _, mg_sd, cr_sd = state_dicts('full')
    _, mgc, crc = oracle_cfgs(FULL)
    _, mg, cr, ph = load_product('full', FULL, dtype=dtype)
    gen = torch.Generator().manual_seed(78)
    ids = torch.randint(0, 65536, (1, 576), generator=gen)
    ctx = weights.synthetic_context(1, 12, 768, seed=1, pad_last=3)
    draws = dict(rand_step=torch.tensor([7]), perm_noise=weights.uniform_noise((1, 576), 710),
                 gumbel_u=weights.uniform_noise((1, 576, 65536), 711))
    leaf = lambda sd: {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith('.beta') else v) for k, v in sd.items()}
    mgl, crl = leaf(mg_sd), leaf(cr_sd)
    ref = O.phenaki_forward_loss(mgl, mgc, crl, crc, ids, patch_shape=(9, 8, 8), context=ctx, steps=FULL['steps'], mask_id=65536, **draws)
    ref['loss'].backward()
    loss = ph(video_codebook_ids=ids.view(1, 9, 8, 8).cuda(), text_embeds=ctx.cuda(), _draws=draws)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref['loss'].detach())) <= 2e-4 * abs(float(ref['loss'].detach()))
    named = [(f'maskgit.{k}', v, mgl[k]) for k, v in mg.named_parameters()] + [(f'critic.{k}', v, crl[k]) for k, v in cr.named_parameters()]
    top = max(float(r.grad.abs().max()) for _, _, r in named if r.grad is not None and r.numel())
    errs = {}
    for name, prm, r in named:
        if r.grad is None or r.numel() == 0:
            assert prm.grad is None or prm.numel() == 0, name
            continue
        if float(r.grad.abs().max()) < 1e-6 * top:
            assert float(prm.grad.abs().max()) <= 1e-2 * tol * top, name
            continue
        errs[name] = close(prm.grad.cpu(), r.grad, tol, f'd {name} ({dtype})')
    worst = max(errs, key=errs.get)
    record_parity('training_step_full_vs_oracle_autograd', dict(dtype=dtype, parameters=len(errs), worst=worst, worst_rel_err=errs[worst],
                                                                median_rel_err=sorted(errs.values())[len(errs) // 2]))


def test_module_forwards_carry_autograd_graphs():
    """MaskGit.forward / TokenCritic.forward / SelfCritic.forward called directly under grad mode (a user's own loss on the logits, as the
    reference allows): values equal the no_grad inference path, gradients equal torch autograd through the oracle"""
    import phenaki_pytorch_amd as P
    cv, mg, cr, ph = load_product('tiny', TINY)
    _, mg_sd, cr_sd = state_dicts('tiny')
    _, mgc, crc = oracle_cfgs(TINY)
    leaf = lambda sd: {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith('.beta') else v) for k, v in sd.items()}
    mgl, crl = leaf(mg_sd), leaf(cr_sd)
    g = torch.Generator().manual_seed(13)
    b, shape = 2, (3, 4, 4)
    ids = torch.randint(0, TINY['maskgit']['num_tokens'] + 1, (b, *shape), generator=g)
    ctx = weights.synthetic_context(b, 6, TINY['maskgit']['dim_context'], seed=4, pad_last=1)
    tmask = torch.any(ctx != 0, dim=-1)
    G = torch.randn(b, 48, TINY['maskgit']['num_tokens'], generator=g)
    ref = O.maskgit_forward(mgl, mgc, ids.flatten(1), video_patch_shape=shape, context=ctx, text_mask=tmask)
    ref.backward(G)
    with torch.no_grad():
        val = mg(ids.cuda(), context=ctx.cuda(), text_mask=tmask.cuda())
    out = mg(ids.cuda(), context=ctx.cuda(), text_mask=tmask.cuda())
    assert out.requires_grad and not val.requires_grad
    close(out.detach().cpu(), ref.detach(), 1e-3, 'MaskGit.forward logits (autograd path)')
    close(out.detach(), val, 1e-3, 'autograd path vs inference path')
    out.backward(G.cuda())
    for name in ('to_logits.weight', 'to_logits.bias', 'token_emb.weight', 'transformer.layers.1.2.to_kv.weight', 'continuous_pos_bias.net.1.0.weight'):
        prm = dict(mg.named_parameters())[name]
        close(prm.grad.cpu(), mgl[name].grad, 1e-3, f'MaskGit.forward d {name}')
    # TokenCritic
    Gc = torch.randn(b, 48, generator=g)
    refc = O.critic_forward(crl, crc, ids.flatten(1).clamp(max=TINY['critic']['num_tokens'] - 1), video_patch_shape=shape, context=ctx, text_mask=tmask)
    refc.backward(Gc)
    cids = ids.clamp(max=TINY['critic']['num_tokens'] - 1).cuda()
    outc = cr(cids, context=ctx.cuda(), text_mask=tmask.cuda())
    assert outc.requires_grad and tuple(outc.shape) == (b, 48)
    close(outc.detach().cpu(), refc.detach(), 1e-3, 'TokenCritic.forward (autograd path)')
    outc.backward(Gc.cuda())
    for name in ('to_logits.0.weight', 'to_logits.0.bias', 'pos_emb.weight', 'transformer.layers.0.3.1.weight'):
        prm = dict(cr.named_parameters())[name]
        close(prm.grad.cpu(), crl[name].grad, 1e-3, f'TokenCritic.forward d {name}')
    # SelfCritic: same trunk as MaskGit + to_pred
    sc = P.SelfCritic(mg).cuda()
    for prm in mg.parameters():
        prm.grad = None
    outs = sc(ids.cuda(), context=ctx.cuda(), text_mask=tmask.cuda())
    assert outs.requires_grad and tuple(outs.shape) == (b, 48)
    outs.sum().backward()
    assert sc.to_pred[0].weight.grad is not None and mg.token_emb.weight.grad is not None


@pytest.mark.parametrize('dtype', ['fp32', 'bf16x3'])
def test_training_loop_tracks_torch_adamw_on_the_oracle(dtype):
    """four optimizer steps of the reference's inner loop (phenaki_trainer.py:351-388: zero_grad, loss, backward, AdamW from optimizer.py) on the
    product against the same loop on the oracle with torch.optim.AdamW on CPU, same draws per step: losses per step and the parameters
    after the last step; then the INFERENCE path must see the trained weights (packed-weight caches are keyed on the tensors' versions,
    which pk_adamw's raw-pointer writes have to bump)"""
    import phenaki_pytorch_amd as P
    cv, mg, cr, ph = load_product('tiny', TINY, dtype=dtype)
    _, mg_sd, cr_sd = state_dicts('tiny')
    _, mgc, crc = oracle_cfgs(TINY)
    leaf = lambda sd: {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith('.beta') else v) for k, v in sd.items()}
    mgl, crl = leaf(mg_sd), leaf(cr_sd)
    b, shape, n = 2, (3, 4, 4), 48
    g = torch.Generator().manual_seed(21)
    ids = torch.randint(0, TINY['maskgit']['num_tokens'], (b, *shape), generator=g)
    ctx = weights.synthetic_context(b, 6, TINY['maskgit']['dim_context'], seed=3, pad_last=2)
    with torch.no_grad():
        before = mg(ids.cuda(), context=ctx.cuda(), text_mask=torch.any(ctx != 0, dim=-1).cuda()).cpu()      # packs the inference weights
    hip_params = [p for p in list(mg.parameters()) + list(cr.parameters())]
    opt = P.get_optimizer(hip_params, lr=2e-3, wd=1e-2)
    ref_params = [v for sd in (mgl, crl) for v in sd.values() if v.requires_grad]
    ref_opt = torch.optim.AdamW([{'params': [p for p in ref_params if p.ndim >= 2]}, {'params': [p for p in ref_params if p.ndim < 2], 'weight_decay': 0}],
                                lr=2e-3, weight_decay=1e-2, betas=(0.9, 0.99), eps=1e-8)
    for it in range(4):
        draws = dict(rand_step=torch.tensor([1 + it, 3]), perm_noise=weights.uniform_noise((b, n), 720 + it),
                     gumbel_u=weights.uniform_noise((b, n, TINY['maskgit']['num_tokens']), 730 + it))
        opt.zero_grad(set_to_none=True)
        loss = ph(video_codebook_ids=ids.cuda(), text_embeds=ctx.cuda(), _draws=draws)
        loss.backward()
        opt.step()
        ref_opt.zero_grad(set_to_none=True)
        ref = O.phenaki_forward_loss(mgl, mgc, crl, crc, ids.flatten(1), patch_shape=shape, context=ctx, steps=TINY['steps'],
                                     mask_id=TINY['maskgit']['num_tokens'], **draws)['loss']
        ref.backward()
        ref_opt.step()
        assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-3 * abs(float(ref.detach())), f'step {it}: {float(loss.detach())} vs {float(ref.detach())}'
    # parameters after 4 steps (Adam turns a noise-level gradient into a full-size +-lr step: the structurally gradient-free bias of the
    # position MLP's last layer is excluded; everything else has a real gradient)
    worst = 0.
    for net, mod, sd in (('maskgit', mg, mgl), ('critic', cr, crl)):
        for k, v in mod.named_parameters():
            if v.numel() == 0 or not sd[k].requires_grad or sd[k].grad is None or k.endswith('continuous_pos_bias.net.2.bias'):
                continue
            got, want_p = v.detach().cpu(), sd[k].detach()
            if dtype == 'fp32':
                worst = max(worst, close(got, want_p, 2e-3, f'{net}.{k} after 4 AdamW steps'))
            else:
                # Adam normalises every element's gradient by its own running magnitude: an element whose gradient sits at the split-bf16 noise
                # floor (1e-5 of the tensor's scale) can take +-lr steps the other way (measured: at most 0.8 lr after 4 steps).  Bound: no
                # element further off than 3 steps' worth, and the tensor as a whole (rms) within 3e-3
                d = (got - want_p)
                assert float(d.abs().max()) <= 3 * 2e-3, f'{net}.{k}: an element is more than three AdamW steps (lr) off'
                rel = float(d.pow(2).mean().sqrt() / want_p.pow(2).mean().sqrt().clamp_min(1e-12))
                assert rel <= 3e-3, f'{net}.{k} after 4 AdamW steps: rms error {rel:.2e}'
                worst = max(worst, rel)
    record_parity('training_loop_4_steps_vs_torch_adamw', dict(dtype=dtype, worst_param_rel_err=worst))
    with torch.no_grad():
        after = mg(ids.cuda(), context=ctx.cuda(), text_mask=torch.any(ctx != 0, dim=-1).cuda()).cpu()
        want = O.maskgit_forward({k: v.detach() for k, v in mgl.items()}, mgc, ids.flatten(1), video_patch_shape=shape, context=ctx,
                                 text_mask=torch.any(ctx != 0, dim=-1))
    assert (after - before).abs().max() > 1e-3, 'training must change the logits'
    close(after, want, 2e-3, 'inference path after training (stale packed weights?)')


@pytest.mark.parametrize('dtype,tol', [('fp32', 1e-5), ('bf16x3', 2e-4), ('bf16', 2e-2)])
def test_gemm_splitk_matches_plain_product(dtype, tol):
    """pk_gemm_splitk + pk_sum_batch (the weight-gradient product of the training step) against an f64 product, ragged M / N tiles"""
    from phenaki_pytorch_amd import _lib as L
    from phenaki_pytorch_amd.attention import resolve_dtype
    from phenaki_pytorch_amd.train import pack_operand, _weight_grad_gemm
    dt = resolve_dtype(dtype)
    g = torch.Generator().manual_seed(31)
    rows, N, K = 1000, 200, 136                     # dW (N, K) = dy^T (N, rows) x (rows, K)
    dy, x = torch.randn(rows, N, generator=g), torch.randn(rows, K, generator=g)
    want = (dy.double().t() @ x.double()).float()
    q = 64 if dtype == 'bf16' else 32
    Mp = (rows + q - 1) // q * q
    dyT = pack_operand(dy.cuda(), dt, transpose=True, side='a')
    xT = pack_operand(x.cuda(), dt, transpose=True)
    for splits in (2, 4):
        if Mp % (splits * q):
            continue
        part = torch.empty((splits, N * K), device='cuda')
        L.gemm_splitk(dt, dyT, xT, N, K, Mp, splits, part)
        out = torch.empty((N, K), device='cuda')
        L.sum_batch(part, splits, out, N * K)
        close(out.cpu(), want, tol, f'split-K x{splits} ({dtype})')
    out2 = torch.empty((N, K), device='cuda')
    _weight_grad_gemm(dt, dyT, xT, N, K, Mp, out2)
    close(out2.cpu(), want, tol, f'weight-gradient product ({dtype})')
    # round 4: 128 x 128 tiles and a bias carried by slice 0 (the split-bf16 patch-embedding product)
    bias = torch.randn(K, generator=g)
    for splits in (2, 4):
        if Mp % (splits * q):
            continue
        part = torch.full((splits, N * K), float('nan'), device='cuda')
        L.gemm_splitk(dt, dyT, xT, N, K, Mp, splits, part, bias=bias.cuda(), tile=1)
        out = torch.empty((N, K), device='cuda')
        L.sum_batch(part, splits, out, N * K)
        close(out.cpu(), want + bias, tol, f'split-K x{splits}, 128-wide tiles + bias ({dtype})')
    # round 6: 256 x 256 tiles on the two-group loop (bf16 only; the other operand types are refused)
    for splits in (2, 4):
        if Mp % (splits * q):
            continue
        part = torch.full((splits, N * K), float('nan'), device='cuda')
        if dtype != 'bf16':
            with pytest.raises(RuntimeError):
                L.gemm_splitk(dt, dyT, xT, N, K, Mp, splits, part, tile=2)
            break
        L.gemm_splitk(dt, dyT, xT, N, K, Mp, splits, part, bias=bias.cuda(), tile=2)
        out = torch.empty((N, K), device='cuda')
        L.sum_batch(part, splits, out, N * K)
        close(out.cpu(), want + bias, tol, f'split-K x{splits}, 256-wide tiles + bias ({dtype})')
