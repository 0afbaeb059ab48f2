"""Per-kernel parity: every C-ABI entry point of libphenaki_hip.so against the CPU oracle / the torch f32
expression of the reference op it replaces.  Needs a real MI355X (-m gpu)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import phenaki_oracle as O
from tests.util import close, uniform24_np, uniform24x4_np

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope='module', autouse=True)
def _oracle_follows_product_ln_fold():
    """the bf16 oracle rounds where the product rounds: LayerNorm folded into the consuming GEMM unless PK_LN_FOLD=0"""
    from phenaki_pytorch_amd import attention
    O.LN_FOLD, O.LN_FOLD_FF, O.LN_FOLD_FF_MAX_ROWS = attention._LN_FOLD, bool(attention._LN_FOLD_FF), attention._LN_FOLD_FF_MAX_ROWS
    O.ATTN_FIXED_OFFSET = attention._ATTN_FIXED
    O.ATTN_FIXED_OFFSET_BIAS = attention._ATTN_FIXED and attention._BIAS_TABLE
    from phenaki_pytorch_amd import cvivit as _cv
    O.PATCH_FUSED = _cv._PATCH_FUSED


@pytest.fixture(scope='module')
def L():
    from phenaki_pytorch_amd import _lib
    _lib.load()
    assert torch.cuda.is_available(), 'these tests need the HIP device (no CPU fallback exists)'
    return _lib


def g(seed):
    gen = torch.Generator().manual_seed(seed)
    return gen


def bf(x):
    return x.to(torch.bfloat16).float()


# ------------------------------------------------------------------------------------------ GEMM

@pytest.mark.parametrize('M,N,K', [(300, 200, 96), (257, 512, 512), (1000, 2048, 512), (64, 8, 64), (77, 2, 128), (130, 1, 64)])
@pytest.mark.parametrize('mode', ['f32', 'bf16', 'bf16_af32'])
def test_gemm_plain_bias_res(L, M, N, K, mode):
    A = torch.randn(M, K, generator=g(1))
    W = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    bias = torch.randn(N, generator=g(3))
    res = torch.randn(M, N, generator=g(4))
    if mode == 'f32':
        dt, Ad, Wd, ref = L.F32, A.cuda(), W.cuda(), A @ W.t() + bias + res
        tol = 2e-5
    else:
        dt = L.BF16
        Wd = W.cuda().to(torch.bfloat16)
        Ad = A.cuda() if mode == 'bf16_af32' else A.cuda().to(torch.bfloat16)
        ref = bf(A) @ bf(W).t() + bias + res
        tol = 2e-5
    C = torch.empty(M, N, device='cuda')
    L.gemm(dt, Ad, Wd, M, N, K, C=C, bias=bias.cuda(), res=res.cuda())
    close(C, ref, tol, f'gemm {mode} {M}x{N}x{K}')


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_gemm_geglu_leaky_gather_and_T_output(L, mode):
    M, K, inner = 333, 128, 344
    A = torch.randn(M, K, generator=g(5))
    W = torch.randn(2 * inner, K, generator=g(6)) / math.sqrt(K)
    dt = L.F32 if mode == 'f32' else L.BF16
    td = L.tdtype(dt)
    cast = (lambda t: t) if mode == 'f32' else bf
    h = cast(A) @ cast(W).t()
    val, gate = h[:, 0::2], h[:, 1::2]                     # interleaved (value, gate) columns
    ref = F.gelu(gate) * val
    C = torch.empty(M, inner, device='cuda', dtype=td)
    L.gemm(dt, A.cuda().to(td), W.cuda().to(td), M, 2 * inner, K, C=C, act=L.ACT_GEGLU)
    close(C.float(), cast(ref) if mode == 'bf16' else ref, 1e-5 if mode == 'f32' else 8e-3, 'geglu')
    # leaky relu + bias, f32 out
    b = torch.randn(2 * inner, generator=g(7))
    C2 = torch.empty(M, 2 * inner, device='cuda')
    L.gemm(dt, A.cuda().to(td), W.cuda().to(td), M, 2 * inner, K, C=C2, bias=b.cuda(), act=L.ACT_LEAKY)
    close(C2, F.leaky_relu(h + b, 0.1), 2e-5, 'leaky')
    # row gather
    idx = torch.randperm(M, generator=g(8))[:100].int()
    C3 = torch.empty(100, 2 * inner, device='cuda')
    L.gemm(dt, A.cuda().to(td), W.cuda().to(td), 100, 2 * inner, K, C=C3, a_rows=idx.cuda())
    close(C3, h[idx.long()], 2e-5, 'gather')


@pytest.mark.parametrize('variant', [0, 1, 2, 8, 9, 24, ('f32only', 3), ('bf16only', 27), ('bf16only', 33), ('bf16only', 50)])
@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('M,N,K', [(300, 200, 96), (1000, 520, 1368), (4608, 512, 512), (129, 2736, 512), (2100, 1160, 64)])
def test_gemm_main_loop_variants(L, variant, mode, M, N, K):
    if isinstance(variant, tuple):
        if (variant[0] == 'bf16only') != (mode == 'bf16'):
            pytest.skip('this main loop is instantiated for the other operand type only')
        variant = variant[1]
    """register-staged and LDS-DMA main loops (tile / ring-depth variants) agree with torch, incl. M/N/K tails, a row
    gather, and the K tail handled by W's zero padding."""
    A = torch.randn(M + 7, K, generator=g(60))
    W = torch.randn(N, K, generator=g(61)) / math.sqrt(K)
    bias = torch.randn(N, generator=g(62))
    res = torch.randn(M, N, generator=g(63))
    idx = torch.randperm(M + 7, generator=g(64))[:M].int()
    dt = L.F32 if mode == 'f32' else L.BF16
    td = L.tdtype(dt)
    cast = (lambda t: t) if mode == 'f32' else bf
    bk = 32 if mode == 'f32' else 64
    Kp = (K + bk - 1) // bk * bk
    Wp = torch.zeros(N, Kp)
    Wp[:, :K] = W
    ref = cast(A)[idx.long()] @ cast(W).t() + bias + res
    C = torch.full((M, N), float('nan'), device='cuda')
    L.gemm(dt, A.cuda().to(td), Wp.cuda().to(td), M, N, K, C=C, bias=bias.cuda(), res=res.cuda(), a_rows=idx.cuda(), variant=variant)
    close(C, ref, 3e-5, f'gemm variant {variant} {mode} {M}x{N}x{K}')
    C2 = torch.full((M, N), float('nan'), device='cuda')
    L.gemm(dt, A.cuda().to(td), Wp.cuda().to(td), M, N, K, C=C2, variant=variant)
    close(C2, cast(A)[:M] @ cast(W).t(), 3e-5, f'gemm variant {variant} {mode} plain')


def test_layernorm_raw_copy_and_transposed_rows(L):
    a, b, c, D = 3, 5, 7, 128
    M = a * b * c
    x = torch.randn(M, D, generator=g(65))
    gamma = 1 + 0.1 * torch.randn(D, generator=g(66))
    ref = F.layer_norm(x, (D,), gamma, None)
    out = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
    raw = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
    out2 = torch.empty(M, D, device='cuda')
    L.layernorm(x.cuda(), gamma.cuda(), None, M, D, out=out, out2=out2, raw=raw, perm=(b, c))
    tr = lambda t: t.view(a, b, c, D).transpose(1, 2).reshape(M, D)
    close(out2, tr(ref), 1e-5, 'perm f32')
    close(out.float(), tr(ref), 5e-3, 'perm bf16')
    assert torch.equal(raw.cpu(), tr(x).to(torch.bfloat16))


def test_gemm_rejects_bad_arguments(L):
    A = torch.randn(8, 10, device='cuda')
    W = torch.randn(8, 10, device='cuda')
    C = torch.empty(8, 8, device='cuda')
    with pytest.raises(RuntimeError, match='PK_EALIGN'):
        L.gemm(L.F32, A, W, 8, 8, 10, C=C)                # K = 10 is not a multiple of 4 f32
    with pytest.raises(RuntimeError, match='PK_EINVAL'):
        L.gemm(L.F32, A, W, 0, 8, 8, C=C)


# ------------------------------------------------------------------------------------------ norms / patch / peg / lfq / embed

@pytest.mark.parametrize('M,D', [(37, 128), (1000, 512), (5, 768), (9, 96), (3, 2048)])
def test_layernorm(L, M, D):
    x = torch.randn(M, D, generator=g(9)) * 3 + 1
    gamma = 1 + 0.1 * torch.randn(D, generator=g(10))
    beta = 0.1 * torch.randn(D, generator=g(11))
    ref = F.layer_norm(x, (D,), gamma, beta)
    out = torch.empty(M, D, device='cuda')
    outb = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
    L.layernorm(x.cuda(), gamma.cuda(), beta.cuda(), M, D, out=outb, out2=out)
    close(out, ref, 1e-5, 'layernorm f32')
    close(outb.float(), ref, 5e-3, 'layernorm bf16')
    out3 = torch.zeros(2 * M + 7, D, device='cuda')
    L.layernorm(x.cuda(), gamma.cuda(), None, M, D, out2=out3[:2 * M], remap=(1, 2, 1))
    close(out3[1:2 * M:2], F.layer_norm(x, (D,), gamma, None), 1e-5, 'layernorm remap')
    assert out3[0:2 * M:2].abs().max().item() == 0


@pytest.mark.parametrize('B,C,Fr,H,W,pt,ph,pw', [(2, 3, 5, 64, 64, 2, 16, 16), (1, 3, 3, 256, 256, 2, 32, 32), (2, 3, 1, 32, 64, 2, 8, 16),
                                                   (1, 3, 3, 128, 128, 2, 32, 16), (1, 3, 3, 64, 64, 2, 32, 4), (2, 1, 5, 64, 256, 2, 32, 64)])
def test_patchify_ln_and_unpatchify(L, B, C, Fr, H, W, pt, ph, pw):
    video = torch.randn(B, C, Fr, H, W, generator=g(12))
    h, w = H // ph, W // pw
    vd = video.cuda()

    def ref_patches(frames, tp):
        t = frames.shape[2] // tp
        return frames.reshape(B, C, t, tp, h, ph, w, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * t * h * w, -1)

    groups = [(0, 1, 1)] + ([(1, (Fr - 1) // pt, pt)] if Fr > 1 else [])
    recon = torch.zeros_like(vd)
    for f0, nt, tp in groups:
        P = C * tp * ph * pw
        wgt = 1 + 0.1 * torch.randn(P, generator=g(13))
        b = 0.1 * torch.randn(P, generator=g(14))
        pat = ref_patches(video[:, :, f0:f0 + nt * tp], tp)
        ref = F.layer_norm(pat, (P,), wgt, b)
        out = torch.empty(B * nt * h * w, P, device='cuda')
        L.patchify_ln(vd, f0, nt, tp, ph, pw, wgt.cuda(), b.cuda(), out)
        close(out, ref, 1e-5, f'patchify_ln f0={f0}')
        outb = torch.empty(B * nt * h * w, P, device='cuda', dtype=torch.bfloat16)
        L.patchify_ln(vd, f0, nt, tp, ph, pw, wgt.cuda(), b.cuda(), outb)
        close(outb.float(), ref, 5e-3, f'patchify_ln bf16 f0={f0}')
        L.unpatchify(pat.cuda().contiguous(), recon, f0, nt, tp, ph, pw)
    assert torch.equal(recon.cpu(), video), 'unpatchify(patchify(video)) must reproduce the video bit-exactly'


@pytest.mark.parametrize('causal', [False, True])
def test_peg(L, causal):
    B, T, H, W, D = 2, 5, 4, 4, 128
    x = torch.randn(B * T * H * W, D, generator=g(15))
    sd = {'dsconv.weight': torch.randn(D, 1, 3, 3, 3, generator=g(16)) * 0.2, 'dsconv.bias': torch.randn(D, generator=g(17)) * 0.1}
    ref = O.peg(sd, '', x.reshape(B, T * H * W, D), (B, T, H, W), causal).reshape(-1, D) + x
    wt = sd['dsconv.weight'].reshape(D, 27).t().contiguous().cuda()
    out = torch.empty_like(x, device='cuda')
    L.peg(x.cuda(), wt, sd['dsconv.bias'].cuda(), out, B, T, H, W, D, causal)
    close(out, ref, 1e-5, 'peg')


def test_lfq_encode_decode(L):
    M, D, cd = 777, 512, 16
    x = torch.randn(M, D, generator=g(18))
    sd = {'vq.project_in.weight': torch.randn(cd, D, generator=g(19)) / math.sqrt(D), 'vq.project_in.bias': torch.randn(cd, generator=g(20)) * 0.05,
          'vq.project_out.weight': torch.randn(D, cd, generator=g(21)) / 4, 'vq.project_out.bias': torch.randn(D, generator=g(22)) * 0.05}
    proj_ref = O.lfq_project(sd, x)
    ids_ref = O.lfq_ids(proj_ref)
    ids = torch.empty(M, device='cuda', dtype=torch.int64)
    proj = torch.empty(M, cd, device='cuda')
    L.lfq_encode(x.cuda(), sd['vq.project_in.weight'].cuda(), sd['vq.project_in.bias'].cuda(), ids, proj, M, D, cd)
    close(proj, proj_ref, 1e-5, 'lfq proj')
    safe = (proj_ref.abs() > 1e-4).all(dim=-1)
    assert torch.equal(ids.cpu()[safe], ids_ref[safe]), 'LFQ ids differ on tokens whose margin exceeds 1e-4'
    assert safe.float().mean() > 0.95
    out = torch.empty(M, D, device='cuda')
    L.lfq_decode(ids_ref.cuda(), sd['vq.project_out.weight'].cuda(), sd['vq.project_out.bias'].cuda(), out, M, D, cd)
    close(out, O.lfq_codes(sd, ids_ref), 1e-5, 'lfq codes')


@pytest.mark.parametrize('M,cd,spread', [(40, 8, 0.02), (300, 16, 0.004), (300, 16, 1.0), (129, 5, 0.01), (64, 2, 0.05), (2000, 12, 0.006)])
def test_lfq_aux_loss_and_gradient_match_oracle(L, M, cd, spread):
    """the LFQ's training-mode auxiliary loss (cvivit.py:570 -> :666; oracle/lfq.py restates the published LFQ.forward, materialising the
    (M, 2^cd) probabilities) from the factorised pk_lfq_aux_* kernels: value, its three parts, and d aux / d project_in(x) against autograd.
    spread: scale of the projections -- ~0.005 keeps alpha z = 400 z around 1 (soft probabilities, the batch-entropy term active), 1.0 saturates
    every bit (probabilities one-hot, everything below the 1e-5 clamp)."""
    from oracle import lfq
    g = torch.Generator().manual_seed(100 * cd + M)
    z = (torch.randn(M, cd, generator=g) * spread).requires_grad_()
    kw = dict(entropy_loss_weight=0.1, commitment_loss_weight=0.25, diversity_gamma=1.)
    bd = {}
    with torch.enable_grad():
        ref = lfq.lfq_aux_loss(z, breakdown=bd, **kw)
        ref.backward()
    out, dz = L.lfq_aux(z.detach().cuda(), **kw)
    out = out.cpu()
    parts = torch.stack([ref.detach(), bd['per_sample_entropy'], bd['codebook_entropy'], bd['commitment']])
    assert (out - parts).abs().max() <= 2e-5 * parts.abs().max(), (out, parts)
    close(dz, z.grad, 2e-4, f'd aux / d proj (M={M}, cd={cd}, spread={spread})')
    # other weights, temperature and scale of the published signature
    kw2 = dict(entropy_loss_weight=0.3, commitment_loss_weight=1.0, diversity_gamma=2.5, inv_temperature=10.)
    z2 = z.detach().clone().requires_grad_()
    with torch.enable_grad():
        ref2 = lfq.lfq_aux_loss(z2, **kw2)
        ref2.backward()
    out2, dz2 = L.lfq_aux(z2.detach().cuda(), **kw2)
    assert abs(float(out2[0]) - float(ref2)) <= 2e-5 * max(1., abs(float(ref2)))
    close(dz2, z2.grad, 2e-4, 'd aux / d proj, non-default weights')


def test_lfq_module_training_forward_returns_the_auxiliary_loss():
    """quantize.LFQ.forward in training mode under grad = the published module's: straight-through codes, ids, differentiable aux (eval: aux = 0)"""
    import phenaki_pytorch_amd as P
    from phenaki_pytorch_amd.quantize import LFQ
    from oracle import lfq
    torch.manual_seed(3)
    m = LFQ(dim=64, codebook_size=256, entropy_loss_weight=0.2, diversity_gamma=1.5).cuda()
    ref = lfq.LFQ(dim=64, codebook_size=256, entropy_loss_weight=0.2, diversity_gamma=1.5)
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    x = (torch.randn(2, 50, 64) * 0.05)
    xr = x.clone().requires_grad_()
    xg = x.cuda().requires_grad_()
    with torch.enable_grad():
        q, ids, aux = m.train()(xg)
        qr, idsr, auxr = ref.train()(xr)
        (q.square().mean() + 3. * aux).backward()
        (qr.square().mean() + 3. * auxr).backward()
    assert torch.equal(ids.cpu(), idsr)
    close(q, qr, 1e-5, 'straight-through codes')
    assert abs(float(aux) - float(auxr)) <= 2e-5 * max(1., abs(float(auxr)))
    close(xg.grad, xr.grad, 5e-4, 'd / d x through project_in (codes + aux)')
    for (k, a), (_, b) in zip(m.named_parameters(), ref.named_parameters()):
        close(a.grad, b.grad, 5e-4, f'd {k}')
    with torch.no_grad():
        q0, ids0, aux0 = m.eval()(x.cuda())
    assert float(aux0) == 0. and torch.equal(ids0.cpu(), idsr)


@pytest.mark.parametrize('M,D,cd', [(4608, 512, 16), (5, 128, 8), (333, 96, 10), (100, 512, 12)])
def test_lfq_decode_shapes(L, M, D, cd):
    """the register-resident outer-product kernel (cd 8 / 16, D/4 dividing 256) and the generic fallback"""
    sd = {'vq.project_in.weight': torch.zeros(cd, D), 'vq.project_out.weight': torch.randn(D, cd, generator=g(21)) / 4,
          'vq.project_out.bias': torch.randn(D, generator=g(22)) * 0.05}
    ids = torch.randint(0, 2 ** cd, (M,), generator=g(23))
    out = torch.full((M, D), float('nan'), device='cuda')
    L.lfq_decode(ids.cuda(), sd['vq.project_out.weight'].cuda(), sd['vq.project_out.bias'].cuda(), out, M, D, cd)
    close(out, O.lfq_codes(sd, ids), 1e-5, f'lfq codes {M}x{D} cd={cd}')


@pytest.mark.parametrize('nb,n_prime,n,pb,pc,D,cd', [(3, 8, 16, 6, 4, 512, 16), (2, 0, 12, 3, 4, 128, 8), (2, 5, 10, 0, 0, 96, 10), (1, 64, 512, 9, 64, 512, 16)])
def test_lfq_decode_prime_ids_and_transposed_rows(L, nb, n_prime, n, pb, pc, D, cd):
    """pk_lfq_decode reading the primed tokens from a second id array (no torch.cat, phenaki_pytorch.py:535-536) and writing its rows
    in the temporal transformer's '(b h w) t' order (cvivit.py:482: no transpose().contiguous() pass) -- bit-identical to the
    concatenate / decode / transpose sequence it replaces."""
    sd = {'vq.project_in.weight': torch.zeros(cd, D), 'vq.project_out.weight': torch.randn(D, cd, generator=g(24)) / 4,
          'vq.project_out.bias': torch.randn(D, generator=g(25)) * 0.05}
    wo, bo = sd['vq.project_out.weight'].cuda(), sd['vq.project_out.bias'].cuda()
    ids = torch.randint(0, 2 ** cd, (nb, n), generator=g(26)).cuda()
    prime = torch.randint(0, 2 ** cd, (nb, n_prime), generator=g(27)).cuda() if n_prime else None
    full = torch.cat((prime, ids), dim=-1) if n_prime else ids
    M = full.numel()
    plain = torch.empty(M, D, device='cuda')
    L.lfq_decode(full.reshape(-1).contiguous(), wo, bo, plain, M, D, cd)
    ref = plain if not pb else plain.view(-1, pb, pc, D).transpose(1, 2).reshape(M, D)
    out = torch.full((M, D), float('nan'), device='cuda')
    L.lfq_decode(ids, wo, bo, out, M, D, cd, ids_prime=prime, perm=(pb, pc))
    assert torch.equal(out, ref)


@pytest.mark.parametrize('a,b,c,D,cd', [(2, 16, 3, 128, 8), (2, 64, 9, 512, 16), (1, 5, 7, 1024, 13), (3, 1, 1, 64, 4)])
def test_layernorm_lfq_fused(L, a, b, c, D, cd):
    """pk_layernorm_lfq = pk_layernorm (rows (a,b,c) -> (a,c,b)) followed by pk_lfq_encode, in one launch"""
    M = a * b * c
    x = torch.randn(M, D, generator=g(30)) * 2 + 0.3
    gamma, beta = 1 + 0.1 * torch.randn(D, generator=g(31)), torch.zeros(D)
    wp, bp = torch.randn(cd, D, generator=g(32)) / math.sqrt(D), torch.randn(cd, generator=g(33)) * 0.05
    y = F.layer_norm(x, (D,), gamma, beta).reshape(a, b, c, D).transpose(1, 2).reshape(M, D)
    proj_ref = y @ wp.t() + bp
    ids_ref = O.lfq_ids(proj_ref)
    ids = torch.full((M,), -1, device='cuda', dtype=torch.int64)
    proj = torch.empty(M, cd, device='cuda')
    tok = torch.empty(M, D, device='cuda')
    L.layernorm_lfq(x.cuda(), gamma.cuda(), beta.cuda(), wp.cuda(), bp.cuda(), ids, M, D, cd, tokens=tok, proj=proj, perm=(b, c))
    close(tok, y, 1e-5, 'fused LN tokens')
    close(proj, proj_ref, 1e-5, 'fused lfq proj')
    safe = (proj_ref.abs() > 1e-4).all(dim=-1)
    assert safe.float().mean() > 0.9
    assert torch.equal(ids.cpu()[safe], ids_ref[safe])
    assert ((ids >= 0) & (ids < 2 ** cd)).all()
    ids2 = torch.full((M,), -1, device='cuda', dtype=torch.int64)           # no optional outputs, identity row order
    L.layernorm_lfq(x.cuda(), gamma.cuda(), None, wp.cuda(), bp.cuda(), ids2, M, D, cd)
    ref2 = O.lfq_ids(F.layer_norm(x, (D,), gamma, None) @ wp.t() + bp)
    p2 = F.layer_norm(x, (D,), gamma, None) @ wp.t() + bp
    safe2 = (p2.abs() > 1e-4).all(dim=-1)
    assert torch.equal(ids2.cpu()[safe2], ref2[safe2])


def test_embed(L):
    V, n, D, S = 257, 20, 128, 3
    tok, pos = torch.randn(V, D, generator=g(23)), torch.randn(64, D, generator=g(24))
    ids = torch.randint(0, V, (S, n), generator=g(25))
    out = torch.empty(S * n, D, device='cuda')
    L.embed(ids.cuda(), tok.cuda(), pos.cuda(), out, S, n, D)
    assert torch.equal(out.cpu(), (tok[ids] + pos[:n]).reshape(S * n, D))
    # cond | null replicas read the same id rows; primed tokens come first (phenaki_pytorch.py:500)
    npr = 7
    prime = torch.randint(0, V, (S, npr), generator=g(26))
    out2 = torch.empty(2 * S * (npr + n), D, device='cuda')
    L.embed(ids.cuda(), tok.cuda(), pos.cuda(), out2, 2 * S, n, D, nb=S, ids_prime=prime.cuda())
    full = torch.cat((prime, ids), dim=1)
    ref2 = (tok[full] + pos[:npr + n]).reshape(S * (npr + n), D)
    assert torch.equal(out2.cpu(), torch.cat((ref2, ref2), dim=0))


@pytest.mark.parametrize('dims,D,heads', [((3, 4, 4), 64, 2), ((8, 8), 128, 8), ((2, 3, 5), 64, 8)])
def test_continuous_position_bias(L, dims, D, heads):
    from phenaki_pytorch_amd.attention import ContinuousPositionBias
    m = ContinuousPositionBias(dim=D, heads=heads, num_dims=len(dims))
    sd = {('cpb.' + k): v for k, v in m.state_dict().items()}
    ref = O.continuous_position_bias(sd, 'cpb.', dims)
    out = m.cuda()(*dims)
    close(out, ref, 1e-4, 'cpb')


# ------------------------------------------------------------------------------------------ attention / transformer

def _attn_module(dim, heads, causal, nnull, dim_context=None, seed=30):
    from phenaki_pytorch_amd.attention import Attention
    torch.manual_seed(seed)
    m = Attention(dim=dim, heads=heads, causal=causal, num_null_kv=nnull, dim_context=dim_context)
    with torch.no_grad():
        m.q_scale.copy_(1 + 0.1 * torch.randn(64))
        m.k_scale.copy_(1 + 0.1 * torch.randn(64))
        m.norm.gamma.copy_(1 + 0.1 * torch.randn(dim))
        if dim_context:
            m.context_norm.gamma.copy_(1 + 0.1 * torch.randn(dim_context))
    return m


@pytest.mark.parametrize('dtype', ['fp32', 'bf16x3', 'bf16'])
@pytest.mark.parametrize('case', ['spatial_bias', 'causal_alibi', 'cross_null_mask', 'self_mask_long'])
def test_attention_block(L, dtype, case):
    from phenaki_pytorch_amd.attention import set_compute_dtype
    dim, heads = 128, 2
    S, n = 3, 64
    ctx = None
    kw = {}
    if case == 'spatial_bias':
        m = _attn_module(dim, heads, False, 0)
        kw['attn_bias'] = torch.randn(heads, n, n, generator=g(31))
    elif case == 'causal_alibi':
        S, n = 7, 9
        m = _attn_module(dim, heads, True, 0)
    elif case == 'cross_null_mask':
        m = _attn_module(dim, heads, False, 2, dim_context=96)
        Lc = 13
        ctx = torch.randn(S, Lc, 96, generator=g(32))
        mask = torch.ones(S, Lc, dtype=torch.bool)
        mask[1, 5:] = False
        mask[2, :] = False                                  # the CFG null branch: only the null keys remain
        kw['mask'] = mask
    else:
        S, n = 2, 200
        m = _attn_module(dim, heads, False, 0)
        mask = torch.rand(S, n, generator=g(33)) > 0.3
        kw['mask'] = mask
        kw['attn_bias'] = torch.randn(heads, n, n, generator=g(34))
    x = torch.randn(S, n, dim, generator=g(35))
    sd = {('a.' + k): v for k, v in m.state_dict().items()}
    ref = O.attention(sd, 'a.', x, heads=heads, causal=m.causal, context=ctx, **kw)
    m = set_compute_dtype(m.cuda(), dtype)
    out = m(x.cuda(), context=ctx.cuda() if ctx is not None else None, mask=kw['mask'].cuda() if 'mask' in kw else None,
            attn_bias=kw['attn_bias'].cuda() if 'attn_bias' in kw else None)
    close(out, ref, {'fp32': 1e-4, 'bf16x3': 2e-4}.get(dtype, 3e-2), f'attention {case} {dtype}')


@pytest.mark.parametrize('S,n,causal,has_bias', [(5, 64, False, True), (23, 9, True, False), (3, 10, True, False), (2, 17, False, True)])
def test_attn_small_equals_prep_plus_flash_path(L, S, n, causal, has_bias):
    """the fused short-sequence kernel (one launch) against the two-kernel MFMA path in exact-f32 mode and the oracle math."""
    h = 2
    q = torch.randn(S * n, h * 64, generator=g(70)).cuda()
    kv = torch.randn(S * n, 2 * h * 64, generator=g(71)).cuda()
    qs = (1 + 0.1 * torch.randn(64, generator=g(72))).cuda()
    ks = (1 + 0.1 * torch.randn(64, generator=g(73))).cuda()
    bias = torch.randn(h, n, n, generator=g(74)).cuda() if has_bias else None
    slopes = torch.tensor([0.5, 0.25]).cuda() if causal else None
    km = (torch.rand(S, n, generator=g(75)) > 0.2)
    km[:, 0] = True
    kmd = km.to(torch.uint8).cuda()
    o1 = torch.empty(S * n, h * 64, device='cuda')
    L.attn_small(q, kv, qs, ks, 8.0, o1, S, h, n, bias=bias, kmask=kmd, slopes=slopes, causal=causal)
    nq_pad, nk_pad = L.attn_pads(n, n, 0)
    Qp = torch.empty(S * h * nq_pad * 64, device='cuda')
    Kp = torch.empty(S * h * nk_pad * 64, device='cuda')
    Vt = torch.empty(S * h * nk_pad * 64, device='cuda')
    L.attn_prep(L.F32, q, kv, None, qs, ks, 8.0, Qp, Kp, Vt, S, h, n, n, 0)
    o2 = torch.empty(S * n, h * 64, device='cuda')
    L.attn_fwd(L.F32, Qp, Kp, Vt, o2, S, h, n, n, 0, bias=bias, kmask=kmd, slopes=slopes, causal=causal)
    close(o1, o2, 1e-5, 'attn_small vs prep+fwd')
    # and the torch expression
    qc, kc, vc = q.cpu().view(S, n, h, 64).transpose(1, 2), kv.cpu()[:, :h * 64].reshape(S, n, h, 64).transpose(1, 2), kv.cpu()[:, h * 64:].reshape(S, n, h, 64).transpose(1, 2)
    sim = torch.einsum('shid,shjd->shij', F.normalize(qc, dim=-1) * qs.cpu(), F.normalize(kc, dim=-1) * ks.cpu()) * 8
    if has_bias:
        sim = sim + bias.cpu()
    sim = sim.masked_fill(~km[:, None, None, :], -torch.finfo(torch.float32).max)
    if causal:
        sim = sim + O.alibi_bias(h, n, n) * 0 - (torch.arange(n)[None, :] - torch.arange(n)[:, None]).abs() * slopes.cpu()[:, None, None]
        sim = sim.masked_fill(torch.ones(n, n, dtype=torch.bool).triu(1), -torch.finfo(torch.float32).max)
    ref = torch.einsum('shij,shjd->shid', sim.softmax(-1), vc).transpose(1, 2).reshape(S * n, h * 64)
    close(o1, ref, 1e-5, 'attn_small vs torch')


@pytest.mark.parametrize('S,n,h,D', [(3, 64, 2, 128), (2, 200, 8, 512), (5, 37, 2, 128)])
def test_qkv_project_writes_the_attention_operand_images(L, S, n, h, D):
    """the fused to_q / to_kv + head split + l2norm + scales + V transpose launch against GEMM -> pk_attn_prep."""
    M = S * n
    xn = torch.randn(M, D, generator=g(80)).to(torch.bfloat16)
    xr = torch.randn(M, D, generator=g(81)).to(torch.bfloat16)
    wq = (torch.randn(h * 64, D, generator=g(82)) / math.sqrt(D)).to(torch.bfloat16)
    wkv = (torch.randn(2 * h * 64, D, generator=g(83)) / math.sqrt(D)).to(torch.bfloat16)
    qs = (1 + 0.1 * torch.randn(64, generator=g(84))).cuda()
    ks = (1 + 0.1 * torch.randn(64, generator=g(85))).cuda()
    nq_pad, nk_pad = L.attn_pads(n, n, 0)
    mk = lambda: torch.zeros(S * h * max(nq_pad, nk_pad) * 64, device='cuda', dtype=torch.bfloat16)
    Qp, Kp, Vt, Qr, Kr, Vr = mk(), mk(), mk(), mk(), mk(), mk()
    L.qkv_project(xn.cuda(), xr.cuda(), wq.cuda(), wkv.cuda(), S, n, h, D, qs, ks, 8.0, Qp, Kp, Vt, nq_pad, nk_pad)
    q = (xn.float() @ wq.float().t()).cuda()
    kv = (xr.float() @ wkv.float().t()).cuda()
    L.attn_prep(L.BF16, q, kv, None, qs, ks, 8.0, Qr, Kr, Vr, S, h, n, n, 0)
    nq_el, nk_el = S * h * nq_pad * 64, S * h * nk_pad * 64
    q4, q4r = Qp[:nq_el].view(S, h, nq_pad, 64)[:, :, :n], Qr[:nq_el].view(S, h, nq_pad, 64)[:, :, :n]
    k4, k4r = Kp[:nk_el].view(S, h, nk_pad, 64)[:, :, :n], Kr[:nk_el].view(S, h, nk_pad, 64)[:, :, :n]
    v4, v4r = Vt[:nk_el].view(S, h, 64, nk_pad)[..., :n], Vr[:nk_el].view(S, h, 64, nk_pad)[..., :n]
    close(q4.float(), q4r.float(), 1e-2, 'Qp')          # one bf16 ulp: the two paths sum the k loop in different orders
    close(k4.float(), k4r.float(), 1e-2, 'Kp')
    close(v4.float(), v4r.float(), 1e-2, 'Vt')
    # query-only form (cross-attention with cached K / V)
    Q2 = mk()
    L.qkv_project(xn.cuda(), None, wq.cuda(), None, S, n, h, D, qs, None, 8.0, Q2, None, None, nq_pad, nk_pad)
    assert torch.equal(Q2[:nq_el].view(S, h, nq_pad, 64)[:, :, :n], q4)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16x3', 'bf16'])
def test_transformer_with_peg_cross_and_ff(L, dtype):
    from phenaki_pytorch_amd.attention import Transformer, set_compute_dtype
    from oracle import weights
    dim, heads, depth = 128, 2, 2
    m = Transformer(dim=dim, depth=depth, heads=heads, dim_context=96, peg=True, has_cross_attn=True)
    weights.fill_module(m, salt=9)
    sd = {('t.' + k): v.clone() for k, v in m.state_dict().items()}
    S, vs = 2, (3, 4, 4)
    n = 48
    x = torch.randn(S, n, dim, generator=g(36))
    ctx = torch.randn(S, 7, 96, generator=g(37))
    cmask = torch.ones(S, 7, dtype=torch.bool)
    cmask[1, 4:] = False
    ref = O.transformer(sd, 't.', x, depth=depth, heads=heads, peg_on=True, cross=True, video_shape=(S, *vs),
                        context=ctx, cross_attn_context_mask=cmask)
    m = set_compute_dtype(m.cuda(), dtype)
    out = m(x.cuda(), video_shape=(S, *vs), context=ctx.cuda(), cross_attn_context_mask=cmask.cuda())
    close(out, ref, {'fp32': 1e-4, 'bf16x3': 2e-4}.get(dtype, 3e-2), f'transformer {dtype}')


# ------------------------------------------------------------------------------------------ sampler kernels

@pytest.mark.parametrize('mode', ['f32', 'bf16x3', 'bf16'])
@pytest.mark.parametrize('M,V,D', [(150, 256, 128), (300, 4096, 512)])
def test_vocab_sample_parity_mode(L, mode, M, V, D):
    e = torch.randn(M, D, generator=g(40))
    W = torch.randn(V, D, generator=g(41)) / math.sqrt(D) * 3
    b = torch.randn(V, generator=g(42)) * 0.1
    U = torch.rand(M, V, generator=g(43))
    dt = {'f32': L.F32, 'bf16': L.BF16, 'bf16x3': L.BF16X3}[mode]
    td = L.tdtype(dt)
    cast = bf if mode == 'bf16' else (lambda t: t)
    logits = cast(e) @ cast(W).t() + b
    T = 0.45
    noisy = logits / T + (-torch.log(-torch.log(U + 1e-10) + 1e-10))
    pred_ref = noisy.argmax(-1)
    score_ref = 1 - logits.softmax(-1).gather(1, pred_ref[:, None]).squeeze(1)
    mask = (torch.rand(M, generator=g(44)) > 0.4)
    ids0 = torch.randint(0, V, (M,), generator=g(45))
    partials = torch.empty(5 * L.vocab_ntiles(V) * M, device='cuda')
    Wd = L.split_planes(W.cuda()) if mode == 'bf16x3' else W.cuda().to(td)
    L.vocab_sample(dt, e.cuda().to(td), Wd, b.cuda(), M, V, D, T, U.cuda(), None, 0, True, partials)
    ids = ids0.clone().cuda()
    pred = torch.empty(M, device='cuda', dtype=torch.int64)
    scores = torch.empty(M, device='cuda')
    L.vocab_reduce(partials, M, V, None, mask.to(torch.uint8).cuda(), ids, pred, scores, True)
    top2 = noisy.topk(2, dim=-1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4 * noisy.abs().max()      # near-ties may flip with the summation order
    assert safe.float().mean() > 0.98
    assert torch.equal(pred.cpu()[safe], pred_ref[safe]), 'gumbel-argmax ids differ outside near-ties'
    same = pred.cpu() == pred_ref
    exp_ids = torch.where(mask, pred.cpu(), ids0)
    assert torch.equal(ids.cpu(), exp_ids)
    exp_scores = torch.where(mask, score_ref, torch.full_like(score_ref, -1e4))
    close(scores.cpu()[same], exp_scores[same], 1e-4, 'confidence scores')


@pytest.mark.parametrize('mode', ['f32', 'bf16x3', 'bf16'])
@pytest.mark.parametrize('M,V,D,with_rows', [(130, 512, 128, False), (77, 1000, 96, True), (300, 65536, 512, True)])
def test_vocab_ce_matches_cross_entropy(L, mode, M, V, D, with_rows):
    """pk_vocab_sample(need_lse) + pk_vocab_ce == F.cross_entropy(reduction='none') of the never-written logits, with and
    without the row indirection (targets live in the full (B*n) array), V tails that do not fill the last 128-wide tile."""
    e = torch.randn(M, D, generator=g(50))
    W = torch.randn(V, D, generator=g(51)) / math.sqrt(D) * 3
    b = torch.randn(V, generator=g(52)) * 0.1
    dt = {'f32': L.F32, 'bf16': L.BF16, 'bf16x3': L.BF16X3}[mode]
    td = L.tdtype(dt)
    cast = bf if mode == 'bf16' else (lambda t: t)
    logits = cast(e) @ cast(W).t() + b
    total = 2 * M + 5
    rows = torch.randperm(total, generator=g(53))[:M].int() if with_rows else None
    targets_full = torch.randint(0, V, (total,), generator=g(54))
    tg = targets_full[rows.long()] if with_rows else targets_full[:M]
    ref = F.cross_entropy(logits, tg, reduction='none')
    q = 64 if mode == 'bf16' else 32
    Wp = torch.zeros(V, (D + q - 1) // q * q)
    Wp[:, :D] = W
    partials = torch.empty(5 * L.vocab_ntiles(V) * M, device='cuda')
    A, Wd = e.cuda().to(td), Wp.cuda().to(td)
    if mode == 'bf16x3':
        Wd = L.split_planes(Wd)
    L.vocab_sample(dt, A, Wd, b.cuda(), M, V, D, 1.0, None, rows.cuda() if with_rows else None, 7, True, partials)
    loss = torch.full((M,), float('nan'), device='cuda')
    L.vocab_ce(dt, partials, M, V, A, Wd, b.cuda(), D, targets_full.cuda(), rows.cuda() if with_rows else None, loss)
    close(loss, ref, {'f32': 2e-5, 'bf16x3': 4e-5}.get(mode, 2e-3), f'vocab_ce {mode} V={V}')


def test_vocab_sample_fast_mode_matches_its_numpy_twin(L):
    M, V, D = 130, 512, 128
    e = torch.randn(M, D, generator=g(46))
    W = torch.randn(V, D, generator=g(47)) / math.sqrt(D) * 3
    b = torch.zeros(V)
    seed = 0x1234567890ABCDEF
    partials = torch.empty(5 * L.vocab_ntiles(V) * M, device='cuda')
    preds = []
    for _ in range(2):
        L.vocab_sample(L.F32, e.cuda(), W.cuda(), b.cuda(), M, V, D, 0.7, None, None, seed, False, partials)
        pred = torch.empty(M, device='cuda', dtype=torch.int64)
        L.vocab_reduce(partials, M, V, None, None, None, pred, None, False)
        preds.append(pred.cpu())
    assert torch.equal(preds[0], preds[1]), 'FAST mode must be deterministic for a fixed seed'
    idx = np.arange(M * V, dtype=np.uint64)
    U = torch.from_numpy(uniform24x4_np(seed, idx)).reshape(M, V)
    assert 0.45 < U.mean() < 0.55 and U.min() >= 0 and U.max() < 1
    for r in range(4):                                           # the four draws of a group are distinct streams
        assert 0.45 < U[:, r::4].mean() < 0.55
    assert abs(np.corrcoef(U[:, 0::4].flatten().numpy(), U[:, 3::4].flatten().numpy())[0, 1]) < 0.02
    noisy = (e @ W.t() + b) / 0.7 + (-torch.log(-torch.log(U)))          # FAST mode has no eps terms: u = 0 -> -inf
    top2 = noisy.topk(2, dim=-1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3 * noisy.abs().max()
    assert safe.float().mean() > 0.95
    assert torch.equal(preds[0][safe], noisy.argmax(-1)[safe])
    L.vocab_sample(L.F32, e.cuda(), W.cuda(), b.cuda(), M, V, D, 0.7, None, None, seed + 1, False, partials)
    pred2 = torch.empty(M, device='cuda', dtype=torch.int64)
    L.vocab_reduce(partials, M, V, None, None, None, pred2, None, False)
    assert (pred2.cpu() != preds[0]).float().mean() > 0.2, 'a different seed must change the draws'
    # a device-resident seed word is ADDED to the by-value seed (hipGraph replays draw fresh noise): seed - 5 + [5] == seed
    sd = torch.tensor([5], device='cuda', dtype=torch.int64)
    L.vocab_sample(L.F32, e.cuda(), W.cuda(), b.cuda(), M, V, D, 0.7, None, None, seed - 5, False, partials, seed_dev=sd)
    pred3 = torch.empty(M, device='cuda', dtype=torch.int64)
    L.vocab_reduce(partials, M, V, None, None, None, pred3, None, False)
    assert torch.equal(pred3.cpu(), preds[0])


@pytest.mark.parametrize('M,V', [(1100, 4136), (1024, 65536), (2500, 1024)])
def test_vocab_head_resident_kernel(L, M, V):
    """bf16, D = 512, M >= 1024 routes pk_vocab_sample to the A-resident persistent kernel (sampler.hip vocab_resident_kernel): rows that do not
    fill the last 128-row panel, a vocabulary that does not fill its last tile / its XCD ranges unevenly (33 tiles over 8 XCDs), workers that
    change row panels.  (1) PARITY noise from memory: gumbel-argmax ids outside near-ties, confidence scores; (2) cross entropy through the
    log-sum-exp partials with the row indirection; (3) FAST noise against the numpy twin of the counter hash, seeded and deterministic."""
    D = 512
    e = torch.randn(M, D, generator=g(60))
    W = torch.randn(V, D, generator=g(61)) / math.sqrt(D) * 3
    b = torch.randn(V, generator=g(62)) * 0.1
    logits = bf(e) @ bf(W).t() + b
    A, Wd, bd = e.cuda().to(torch.bfloat16), W.cuda().to(torch.bfloat16), b.cuda()
    partials = torch.empty(5 * L.vocab_ntiles(V) * M, device='cuda')

    def reduce(need_lse, rows=None, mask=None, ids=None):
        pred = torch.empty(M if rows is None else int(rows.max()) + 1, device='cuda', dtype=torch.int64)
        scores = torch.empty_like(pred, dtype=torch.float32) if need_lse else None
        L.vocab_reduce(partials, M, V, rows, mask, ids, pred, scores, need_lse)
        return pred.cpu(), (scores.cpu() if need_lse else None)
    # (1) PARITY with explicit uniforms (memory-bound on the noise: only for the two smaller vocabularies)
    if M * V <= 5_000_000:
        U = torch.rand(M, V, generator=g(63))
        T = 0.45
        noisy = logits / T + (-torch.log(-torch.log(U + 1e-10) + 1e-10))
        partials.fill_(float('nan'))
        L.vocab_sample(L.BF16, A, Wd, bd, M, V, D, T, U.cuda(), None, 0, True, partials)
        pred, scores = reduce(True)
        top2 = noisy.topk(2, dim=-1).values
        safe = (top2[:, 0] - top2[:, 1]) > 1e-4 * noisy.abs().max()
        assert safe.float().mean() > 0.98 and torch.equal(pred[safe], noisy.argmax(-1)[safe])
        same = pred == noisy.argmax(-1)
        score_ref = 1 - logits.softmax(-1).gather(1, noisy.argmax(-1)[:, None]).squeeze(1)
        close(scores[same], score_ref[same], 1e-4, 'confidence scores (resident kernel)')
    # (2) cross entropy with a row indirection
    total = 2 * M + 5
    rows = torch.randperm(total, generator=g(64))[:M].int()
    targets_full = torch.randint(0, V, (total,), generator=g(65))
    partials.fill_(float('nan'))
    L.vocab_sample(L.BF16, A, Wd, bd, M, V, D, 1.0, None, rows.cuda(), 7, True, partials, no_noise=True)
    loss = torch.full((M,), float('nan'), device='cuda')
    L.vocab_ce(L.BF16, partials, M, V, A, Wd, bd, D, targets_full.cuda(), rows.cuda(), loss)
    close(loss, F.cross_entropy(logits, targets_full[rows.long()], reduction='none'), 2e-3, f'vocab_ce, resident kernel, V={V}')
    pred_plain, _ = reduce(False)
    top2 = logits.topk(2, dim=-1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4 * logits.abs().max()
    assert torch.equal(pred_plain[:M][safe], logits.argmax(-1)[safe]), 'plain argmax (no_noise)'
    # (3) FAST noise: the numpy twin of the hash, indexed by (logical row, column)
    if M * V <= 5_000_000:
        seed = 0x0FEDCBA987654321
        outs = []
        for _ in range(2):
            partials.fill_(float('nan'))
            L.vocab_sample(L.BF16, A, Wd, bd, M, V, D, 0.7, None, None, seed, False, partials)
            outs.append(reduce(False)[0])
        assert torch.equal(outs[0], outs[1])
        Uf = torch.from_numpy(uniform24x4_np(seed, np.arange(M * V, dtype=np.uint64))).reshape(M, V)
        noisy = logits / 0.7 + (-torch.log(-torch.log(Uf)))
        top2 = noisy.topk(2, dim=-1).values
        safe = (top2[:, 0] - top2[:, 1]) > 1e-3 * noisy.abs().max()
        assert safe.float().mean() > 0.95 and torch.equal(outs[0][safe], noisy.argmax(-1)[safe])


@pytest.mark.parametrize('B,n,k', [(3, 48, 1), (3, 48, 47), (2, 576, 288), (1, 1024, 50)])
def test_topk_mask(L, B, n, k):
    scores = torch.randn(B, n, generator=g(48))
    ids0 = torch.randint(0, 100, (B, n), generator=g(49))
    idx = scores.topk(k, dim=-1).indices
    mask_ref = torch.zeros(B, n).scatter(1, idx, 1).bool()
    mask = torch.zeros(B, n, device='cuda', dtype=torch.uint8)
    ids = ids0.clone().cuda()
    rows = torch.full((B * k,), -1, device='cuda', dtype=torch.int32)
    nxt = torch.zeros(B, n, device='cuda')
    L.topk_mask(scores.cuda(), B, n, k, 100, mask, ids, rows, scores_next=nxt)
    assert (nxt == -1e4).all()
    assert torch.equal(mask.cpu().bool(), mask_ref)
    assert torch.equal(ids.cpu(), torch.where(mask_ref, 100, ids0))
    flat = (idx + torch.arange(B)[:, None] * n).reshape(-1)            # topk order = descending score = rank order
    assert torch.equal(rows.cpu().long(), flat)


def test_cfg_mix_and_critic_head(L):
    nb, n_tot, n_prime, D = 2, 10, 3, 128
    x = torch.randn(2 * nb * n_tot, D, generator=g(50))
    xc, xn = x[:nb * n_tot].reshape(nb, n_tot, D), x[nb * n_tot:].reshape(nb, n_tot, D)
    ref = (xn + (xc - xn) * 5.)[:, n_prime:].reshape(-1, D)
    out = torch.empty(nb * (n_tot - n_prime), D, device='cuda')
    L.cfg_mix(x.cuda(), nb, n_tot, n_prime, None, nb * (n_tot - n_prime), 5., True, out, D)
    close(out, ref, 1e-6, 'cfg mix')
    w, b = torch.randn(D, generator=g(51)), torch.randn(1, generator=g(52))
    u = torch.rand(nb, n_tot - n_prime, generator=g(53))
    sc, sn = xc @ w + b, xn @ w + b
    ref_s = (sn + (sc - sn) * 5.)[:, n_prime:] + 0.5 * (u - 0.5)
    outs = torch.empty(nb, n_tot - n_prime, device='cuda')
    L.critic_head(x.cuda(), w.cuda(), b.cuda(), D, nb, n_tot, n_prime, True, 5., u.cuda(), 0.5, outs)
    close(outs, ref_s, 1e-5, 'critic head')
    # FAST mode: the uniform draw comes from the counter hash (seed by value + device word), stream 0xC817
    seed = 0x0123456789ABCDEF
    sd = torch.tensor([1000], device='cuda', dtype=torch.int64)
    L.critic_head(x.cuda(), w.cuda(), b.cuda(), D, nb, n_tot, n_prime, True, 5., None, 0.5, outs, seed=seed - 1000, seed_dev=sd)
    r = np.arange(nb * (n_tot - n_prime), dtype=np.uint64) | (np.uint64(0xC817) << np.uint64(32))
    uh = torch.from_numpy(uniform24_np(seed, r)).reshape(nb, n_tot - n_prime)
    close(outs, (sn + (sc - sn) * 5.)[:, n_prime:] + 0.5 * (uh - 0.5), 1e-5, 'critic head hash noise')


@pytest.mark.parametrize('S,n,causal,with_bias,heads', [(18, 64, False, True, 8), (130, 9, True, False, 8), (7, 10, True, False, 2),
                                                        (5, 64, False, False, 8), (3, 48, False, True, 2), (9, 7, False, True, 4),
                                                        (1, 1, True, False, 2), (2, 64, True, False, 2)])
@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
def test_qkv_attn_fused_short_sequences(L, S, n, causal, with_bias, heads, mode):
    """pk_qkv_attn (projections + attention of whole short sequences in one launch) against the oracle in bf16 precision, and
    against the two-launch path it replaces (pk_qkv_project + pk_attn_fwd), through the product Attention module."""
    from phenaki_pytorch_amd import attention as A
    D = 128
    torch.manual_seed(100 + n)
    att = A.Attention(dim=D, heads=heads, causal=causal)
    with torch.no_grad():
        att.q_scale.add_(0.1 * torch.randn(64))
        att.k_scale.add_(0.1 * torch.randn(64))
        att.norm.gamma.add_(0.1 * torch.randn(D))
    sd = {'a.' + k: v.detach().clone() for k, v in att.state_dict().items()}
    att = att.cuda().eval()
    x = torch.randn(S, n, D, generator=g(7 + n)) * 1.5 + 0.2
    bias = torch.randn(heads, n, n, generator=g(8 + n)) if with_bias else None
    dt = _MODE_DT(L, mode)
    with O.precision(mode):
        ref = O.attention(sd, 'a.', x, heads=heads, causal=causal, attn_bias=bias)
    ref_f32 = O.attention(sd, 'a.', x, heads=heads, causal=causal, attn_bias=bias)
    xg = x.reshape(S * n, D).cuda()
    bg = bias.cuda() if with_bias else None
    assert A._SHORT_FUSED and A.ln_fold_enabled(dt)
    fused = att.run(xg, S, n, dt, attn_bias=bg) - xg
    A._SHORT_FUSED = False
    try:
        split = att.run(xg, S, n, dt, attn_bias=bg) - xg          # pk_qkv_project + pk_attn_fwd (either mode, round 4)
    finally:
        A._SHORT_FUSED = True
    gap = ((ref - ref_f32).abs().max() / ref_f32.abs().max()).item()
    tol = 3e-3 if mode == 'bf16' else 1e-4                        # split-bf16: the f32 oracle at f32-grade tolerance
    close(fused.view(S, n, D), ref, tol, f'fused qkv+attn {mode} S={S} n={n} (bf16-vs-f32 gap {gap:.1e})')
    close(fused, split, tol, 'fused vs pk_qkv_project + pk_attn_fwd')
    assert torch.isfinite(fused).all()
    if mode == 'bf16x3':
        # and against the round-3 path of the mode: separate LayerNorm, q / kv GEMMs, pk_attn_prep, LDS-free attention
        A._LN_FOLD_X3 = False
        try:
            unfolded = att.run(xg, S, n, dt, attn_bias=bg) - xg
        finally:
            A._LN_FOLD_X3 = True
        close(fused, unfolded, 1e-4, 'folded + fused vs the unfolded split-bf16 path')


_MODE_DT = lambda L, mode: {'f32': L.F32, 'bf16': L.BF16, 'bf16x3': L.BF16X3}[mode]       # noqa: E731


@pytest.mark.parametrize('mode', ['f32', 'bf16', 'bf16x3'])
@pytest.mark.parametrize('M,N,K,geglu', [(300, 192, 128, False), (1000, 512, 512, False), (700, 2736, 512, True), (129, 64, 1368, False)])
def test_gemm_layernorm_fold_and_bf16_copy(L, mode, M, N, K, geglu):
    """pk_gemm_ex with ln_s / ln_t: LN(x) W^T = rstd * (x (gamma.W)^T - mean * s) + t from the A tiles' own statistics, against
    the same expression in torch (operands rounded as the MFMAs see them) and against the plain LayerNorm + Linear; C2 = the
    bf16 copy of an f32 result (the next block's operand)."""
    from torch import nn
    from phenaki_pytorch_amd import attention as A
    dt = _MODE_DT(L, mode)
    td = L.tdtype(dt)
    cast = bf if mode == 'bf16' else (lambda t: t)              # split-bf16 (round 4: the fold exists for it) is held to the f32 expression
    x = torch.randn(M, K, generator=g(70)) * 1.3 + 0.4
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g(71)), 0.1 * torch.randn(K, generator=g(72))
    W = torch.randn(N, K, generator=g(73)) / math.sqrt(K)
    owner = nn.Linear(1, 1)
    Wd, gd, bd = W.cuda(), gamma.cuda(), beta.cuda()
    wg, s, t, beta_zero = A.folded_weight(owner, 'k', lambda: Wd, gd, bd, dt, [gd])
    assert not beta_zero and wg.dtype == td and wg.shape[0] == N
    xb = cast(x)
    mean = xb.mean(-1, keepdim=True)
    rstd = 1 / torch.sqrt((xb * xb).mean(-1, keepdim=True) - mean * mean + 1e-5)
    wgr = cast(W * gamma)
    ref = rstd * (xb @ wgr.t() - mean * wgr.sum(-1)) + W @ beta
    true = F.layer_norm(x, (K,), gamma, beta) @ W.t()
    close(ref, true, 3e-2 if mode == 'bf16' else 1e-4, 'folded expression vs LayerNorm + Linear')
    if geglu:
        ref = F.gelu(ref[:, 1::2]) * ref[:, 0::2]
        C = torch.full((M, N // 2), float('nan'), device='cuda', dtype=td)
        L.gemm(dt, x.cuda().to(td), wg, M, N, K, C=C, act=L.ACT_GEGLU, ln=(s, t, 1e-5))
        close(C.float(), cast(ref), {'f32': 2e-5, 'bf16x3': 1e-4, 'bf16': 8e-3}[mode], f'ln-folded geglu gemm {mode}')
        return
    res = torch.randn(M, N, generator=g(74))
    C = torch.full((M, N), float('nan'), device='cuda')
    C2 = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16) if mode == 'bf16' else None
    L.gemm(dt, x.cuda().to(td), wg, M, N, K, C=C, res=res.cuda(), ln=(s, t, 1e-5), C2=C2)
    close(C, ref + res, 1e-4 if mode == 'bf16x3' else 3e-5, f'ln-folded gemm {mode} {M}x{N}x{K}')
    if C2 is not None:
        assert torch.equal(C2.float(), bf(C.cpu()).cuda()), 'C2 must be the bf16 rounding of C'
    # plain GEMM with a bf16 copy (the to_out / FF2 producers)
    if mode == 'bf16':
        C3 = torch.empty(M, N, device='cuda')
        C4 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        L.gemm(dt, x.cuda().to(td), pack := A.pack_linear_weight(Wd, dt), M, N, K, C=C3, res=res.cuda(), C2=C4)
        close(C3, xb @ bf(W).t() + res, 3e-5, 'gemm + C2')
        assert torch.equal(C4.float(), bf(C3.cpu()).cuda())


@pytest.mark.parametrize('mode', ['f32', 'bf16', 'bf16x3'])
@pytest.mark.parametrize('M,N,K,D,variant', [(200, 304, 64, 96, 0), (4608, 1368 * 2, 512, 512, 0), (700, 96, 2048, 72, 0), (9216, 64, 128, 512, 24)])
def test_gemm_row_stats_handoff(L, mode, M, N, K, D, variant):
    """stats_out -> ln_stats: the producer (x = o Wo^T + res, D columns) leaves per-32-column (sum, sum of squares) of the rows as its
    consumer reads them; the LayerNorm-folded consumer (N columns over K = D) must match the same GEMM with in-loop statistics."""
    from torch import nn
    from phenaki_pytorch_amd import attention as A
    dt = _MODE_DT(L, mode)
    td = L.tdtype(dt)
    cast = bf if mode == 'bf16' else (lambda t: t)
    lo = {'f32': 3e-5, 'bf16x3': 1e-4, 'bf16': 8e-3}[mode]
    o = torch.randn(M, K, generator=g(90))
    Wo = torch.randn(D, K, generator=g(91)) / math.sqrt(K)
    res = torch.randn(M, D, generator=g(92)) * 2 + 0.3
    x = torch.full((M, D), float('nan'), device='cuda')
    xt = torch.full((M, D), float('nan'), device='cuda', dtype=torch.bfloat16) if mode == 'bf16' else None
    npart = (D + 31) // 32
    stats = torch.full((M, npart, 2), float('nan'), device='cuda')
    L.gemm(dt, o.cuda().to(td), A.pack_linear_weight(Wo.cuda(), dt), M, D, K, C=x, res=res.cuda(), C2=xt, stats_out=stats, variant=variant)
    close(x, cast(o) @ cast(Wo).t() + res, 1e-4 if mode == 'bf16x3' else 3e-5, 'producer')
    seen = (xt.float() if xt is not None else x).cpu().double()
    pad = torch.zeros(M, npart * 32, dtype=torch.float64)
    pad[:, :D] = seen
    pad = pad.view(M, npart, 32)
    close(stats[..., 0], pad.sum(-1).float(), 1e-5, 'stats: sums')
    close(stats[..., 1], (pad * pad).sum(-1).float(), 1e-5, 'stats: sums of squares')
    # consumer
    gamma, beta = 1 + 0.2 * torch.randn(D, generator=g(93)), 0.1 * torch.randn(D, generator=g(94))
    W = torch.randn(N, D, generator=g(95)) / math.sqrt(D)
    Wd, gd, bd = W.cuda(), gamma.cuda(), beta.cuda()
    wg, s, t, _ = A.folded_weight(nn.Linear(1, 1), 'k', lambda: Wd, gd, bd, dt, [gd])
    a = xt if xt is not None else x
    for act in (L.ACT_NONE, L.ACT_GEGLU):
        shape = (M, N // 2) if act == L.ACT_GEGLU else (M, N)
        c1 = torch.full(shape, float('nan'), device='cuda', dtype=td)
        c2 = torch.full(shape, float('nan'), device='cuda', dtype=td)
        L.gemm(dt, a, wg, M, N, D, C=c1, act=act, ln=(s, t, 1e-5))
        L.gemm(dt, a, wg, M, N, D, C=c2, act=act, ln=(s, t, 1e-5), ln_stats=stats)
        close(c2.float(), c1.float(), {'f32': 1e-5, 'bf16x3': 3e-5, 'bf16': 8e-3}[mode], f'handed-over vs in-loop statistics, act {act}')
        xs = seen.float()
        mean = xs.mean(-1, keepdim=True)
        rstd = 1 / torch.sqrt((xs * xs).mean(-1, keepdim=True) - mean * mean + 1e-5)
        wgr = cast(W * gamma)
        ref = rstd * (xs @ wgr.t() - mean * wgr.sum(-1)) + W @ beta
        if act == L.ACT_GEGLU:
            ref = F.gelu(ref[:, 1::2]) * ref[:, 0::2]
        close(c2.float(), cast(ref), lo, f'ln_stats gemm {mode} act {act}')


@pytest.mark.parametrize('mode', ['f32', 'bf16', 'bf16x3'])
@pytest.mark.parametrize('M,N,K,variant', [(200, 96, 64, 0), (4608, 512, 512, 0), (9216, 512, 512, 27), (333, 128, 192, 24)])
def test_gemm_dup_rows(L, mode, M, N, K, variant):
    """pk_gemm_ex dup_rows: every output row of C (and of its bf16 copy C2) is written a second time dup_rows rows further down -- the
    cond | null copies of a CFG batch from ONE to_out GEMM; rows outside both copies stay untouched."""
    from phenaki_pytorch_amd import attention as A
    if mode == 'f32' and variant in (27,):
        pytest.skip('bf16-only tile')
    dt = _MODE_DT(L, mode)
    td = L.tdtype(dt)
    cast = bf if mode == 'bf16' else (lambda t: t)
    x = torch.randn(M, K, generator=g(120))
    W = torch.randn(N, K, generator=g(121)) / math.sqrt(K)
    res = torch.randn(M, N, generator=g(122))
    gap = 5                                                     # the second copy need not be adjacent
    C = torch.full((2 * M + gap, N), float('nan'), device='cuda')
    C2 = torch.full((2 * M + gap, N), float('nan'), device='cuda', dtype=torch.bfloat16) if mode == 'bf16' else None
    L.gemm(dt, x.cuda().to(td), A.pack_linear_weight(W.cuda(), dt), M, N, K, C=C, res=res.cuda(), C2=C2, dup_rows=M + gap, variant=variant)
    ref = cast(x) @ cast(W).t() + res
    close(C[:M], ref, 1e-4 if mode == 'bf16x3' else 3e-5, 'first copy')
    assert torch.equal(C[M + gap:], C[:M]), 'the second copy must be bit-identical'
    assert torch.isnan(C[M:M + gap]).all()
    if C2 is not None:
        assert torch.equal(C2[:M].float(), bf(C[:M].cpu()).cuda()) and torch.equal(C2[M + gap:], C2[:M]) and torch.isnan(C2[M:M + gap].float()).all()
    with pytest.raises(RuntimeError):                            # a second copy that would overlap the first is refused
        L.gemm(dt, x.cuda().to(td), A.pack_linear_weight(W.cuda(), dt), M, N, K, C=C, res=res.cuda(), dup_rows=M - 1, variant=variant)


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('B,T,pt,hw,p,variant', [(2, 3, 2, 4, 8, 0), (1, 2, 2, 8, 4, 0), (3, 1, 1, 2, 8, 1), (2, 3, 2, 4, 8, 24)])
def test_gemm_scatter_epilogue_unpatchify(L, mode, B, T, pt, hw, p, variant):
    """pk_gemm_ex row_off / col_off: the gathered to_pixels GEMM writes 'b t h w (c pt p1 p2) -> b c (t pt) (h p1) (w p2)' in place
    (cvivit.py:326-334, 505-516); checked against the (rows, P) product rearranged by torch, untouched frames stay NaN."""
    from phenaki_pytorch_amd.cvivit import CViViT
    dt = L.F32 if mode == 'f32' else L.BF16
    td = L.tdtype(dt)
    cast = (lambda t: t) if mode == 'f32' else bf
    C, D = 3, 64
    net = CViViT(dim=D, codebook_size=16, image_size=hw * p, patch_size=p, temporal_patch_size=pt, spatial_depth=1, temporal_depth=1,
                 dim_head=64, heads=1, channels=C)
    Fr = 1 + (T - 1) * pt
    x = torch.randn(B * T * hw * hw, D, generator=g(80))
    video = torch.full((B, C, Fr, hw * p, hw * p), float('nan'), device='cuda')
    for f0, ntg, ptg, row0 in ((0, 1, 1, 0), (1, T - 1, pt, hw * hw)):
        if ntg <= 0:
            continue
        P = C * ptg * p * p
        W = torch.randn(P, D, generator=g(81 + ptg)) / math.sqrt(D)
        bias = torch.randn(P, generator=g(83))
        idx, row_off, col_off = net._unpatch_maps(B, T, f0, ntg, ptg, row0, torch.device('cuda'))
        L.gemm(dt, x.cuda().to(td), W.cuda().to(td), idx.numel(), P, D, C=video, bias=bias.cuda(), a_rows=idx,
               scatter=(row_off, col_off), variant=variant)
        rows = x.view(B, T, hw * hw, D)[:, (0 if f0 == 0 else 1):(1 if f0 == 0 else T)].reshape(-1, D)
        pix = cast(rows) @ cast(W).t() + bias
        ref = pix.view(B, ntg, hw, hw, C, ptg, p, p).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(B, C, ntg * ptg, hw * p, hw * p)
        close(video[:, :, f0:f0 + ntg * ptg], ref, 3e-5, f'scatter gemm {mode} frames {f0}+{ntg * ptg}')
        if f0 == 0 and T > 1:
            assert torch.isnan(video[:, :, 1:]).all(), 'first-frame group must not touch the other frames'
    assert not torch.isnan(video).any()


def test_peg_and_embed_bf16_copies(L):
    B, T, H, W, D = 2, 3, 4, 8, 64
    x = torch.randn(B * T * H * W, D, generator=g(80))
    wt, bias = torch.randn(27, D, generator=g(81)) * 0.1, torch.randn(D, generator=g(82)) * 0.1
    out, out_t = torch.empty_like(x, device='cuda'), torch.empty(x.shape, device='cuda', dtype=torch.bfloat16)
    L.peg(x.cuda(), wt.cuda(), bias.cuda(), out, B, T, H, W, D, True, out_t=out_t)
    out2 = torch.empty_like(x, device='cuda')
    L.peg(x.cuda(), wt.cuda(), bias.cuda(), out2, B, T, H, W, D, True)
    assert torch.equal(out, out2) and torch.equal(out_t.float(), bf(out.cpu()).cuda())
    V, n, S = 50, 12, 3
    tok, pos = torch.randn(V, D, generator=g(83)), torch.randn(32, D, generator=g(84))
    ids = torch.randint(0, V, (S, n), generator=g(85))
    e, e_t = torch.empty(S * n, D, device='cuda'), torch.empty(S * n, D, device='cuda', dtype=torch.bfloat16)
    L.embed(ids.cuda(), tok.cuda(), pos.cuda(), e, S, n, D, out_t=e_t)
    assert torch.equal(e.cpu(), (tok[ids] + pos[:n]).reshape(S * n, D)) and torch.equal(e_t.float(), bf(e.cpu()).cuda())


@pytest.mark.parametrize('S,n,n_ctx,masked', [(3, 64, 13, True), (2, 192, 40, True), (2, 128, 62, False), (4, 64, 1, False)])
@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
def test_cross_attention_cached_fused(L, S, n, n_ctx, masked, mode):
    """pk_q_attn_cached (query projection + attention against the cached context K / V in one launch) through the product
    Attention module: first call fills the cache (pk_attn_prep path), second call takes the fused kernel; both against the oracle
    in bf16 precision, incl. the CFG null branch (every text key masked: only the null keys remain)."""
    from phenaki_pytorch_amd import attention as A
    D, heads, dc = 128, 2, 96
    torch.manual_seed(200 + n_ctx)
    att = A.Attention(dim=D, heads=heads, num_null_kv=2, dim_context=dc)
    with torch.no_grad():
        att.q_scale.add_(0.1 * torch.randn(64))
        att.k_scale.add_(0.1 * torch.randn(64))
        att.norm.gamma.add_(0.1 * torch.randn(D))
        att.context_norm.gamma.add_(0.1 * torch.randn(dc))
    sd = {'a.' + k: v.detach().clone() for k, v in att.state_dict().items()}
    att = att.cuda().eval()
    x = torch.randn(S, n, D, generator=g(9 + n)) * 1.5 + 0.2
    ctx = torch.randn(S, n_ctx, dc, generator=g(10 + n))
    mask = torch.ones(S, n_ctx, dtype=torch.bool)
    if masked:
        mask[1, n_ctx // 2:] = False
        mask[-1, :] = False
    dt = _MODE_DT(L, mode)
    tol = 3e-3 if mode == 'bf16' else 1e-4
    with O.precision(mode):
        ref = O.attention(sd, 'a.', x, heads=heads, context=ctx, mask=mask)
    xg = x.reshape(S * n, D).cuda()
    cache = {}
    kw = dict(context2d=ctx.reshape(-1, dc).cuda(), n_ctx=n_ctx, kmask=mask.to(torch.uint8).cuda(), kv_cache=cache)
    first = att.run(xg, S, n, dt, **kw) - xg
    assert len(cache) == 1
    fused = att.run(xg, S, n, dt, **kw) - xg
    close(first.view(S, n, D), ref, tol, 'cross-attention, cache-filling call')
    close(fused.view(S, n, D), ref, tol, f'fused cached cross-attention {mode} n={n} n_ctx={n_ctx}')
    close(fused, first, tol, 'fused vs unfused')


@pytest.mark.parametrize('dims,S,heads', [((9, 8, 8), 2, 8), ((3, 8, 8), 3, 2), ((10, 8, 8), 1, 8), ((4, 5, 4), 2, 2), ((5, 5, 5), 1, 2), ((2, 9, 12), 2, 1)])
def test_attention_relative_position_bias_table(L, dims, S, heads):
    """the continuous position bias as a relative-position TABLE in LDS (pk_attn_fwd bias_tab) against the full (heads, n, n)
    matrix streamed from memory: the table holds the matrix' own entries, so the attention outputs must be identical."""
    from phenaki_pytorch_amd.attention import ContinuousPositionBias
    n = dims[0] * dims[1] * dims[2]
    torch.manual_seed(7)
    cpb = ContinuousPositionBias(dim=64, heads=heads, num_dims=3).cuda()
    full = cpb(*dims)
    tab, codes, off, lo, hi, run4 = cpb.table(*dims)
    assert run4 == (dims[-1] % 4 == 0)
    assert lo == full.min().item() and hi == full.max().item()
    assert tuple(full.shape) == (heads, n, n) and tab.shape[0] == heads and codes.dtype == torch.int32
    idx = (codes.long()[:, None] - codes.long()[None, :] + off)
    assert idx.min() >= 0 and idx.max() < tab.shape[1]
    assert torch.equal(tab[:, idx], full), 'the table must reproduce every entry of the bias matrix bit for bit'
    nq_pad, nk_pad = L.attn_pads(n, n, 0)
    Qp = (torch.randn(S * heads * nq_pad * 64, generator=g(91)) * 0.35).cuda().to(torch.bfloat16)
    Kp = (torch.randn(S * heads * nk_pad * 64, generator=g(92)) * 0.35).cuda().to(torch.bfloat16)
    Vt = torch.randn(S * heads * nk_pad * 64, generator=g(93)).cuda().to(torch.bfloat16)
    o_full = torch.empty(S * n, heads * 64, device='cuda', dtype=torch.bfloat16)
    o_tab = torch.full_like(o_full, float('nan'))
    L.attn_fwd(L.BF16, Qp, Kp, Vt, o_full, S, heads, n, n, 0, bias=full)
    L.attn_fwd(L.BF16, Qp, Kp, Vt, o_tab, S, heads, n, n, 0, bias_table=(tab, codes, off))
    assert torch.isfinite(o_tab.float()).all()
    close(o_tab.float(), o_full.float(), 1e-6 if n % 64 == 0 else 4e-3, f'table vs matrix bias {dims}')
    # fixed-offset softmax (score_bound) against the running-max loop and against an f64 softmax of the same operand images
    Q = Qp.view(S * heads, nq_pad, 64)[:, :n].double().cpu()
    K = Kp.view(S * heads, nk_pad, 64)[:, :n].double().cpu()
    V = Vt.view(S * heads, 64, nk_pad)[:, :, :n].double().cpu()
    bias_d = full.double().cpu().repeat(S, 1, 1)
    for what, kw, extra in (('table', dict(bias_table=(tab, codes, off)), bias_d), ('table, 4-key runs', dict(bias_table=cpb.table(*dims)), bias_d),
                            ('no bias', {}, 0.)):
        sim = Q @ K.transpose(1, 2) + extra
        ref = (sim.softmax(-1) @ V.transpose(1, 2)).view(S, heads, n, 64).permute(0, 2, 1, 3).reshape(S * n, heads * 64)
        bound = float(sim.max()) + 0.3
        o_fix = torch.full_like(o_full, float('nan'))
        o_run = torch.full_like(o_full, float('nan'))
        L.attn_fwd(L.BF16, Qp, Kp, Vt, o_run, S, heads, n, n, 0, **kw)
        L.attn_fwd(L.BF16, Qp, Kp, Vt, o_fix, S, heads, n, n, 0, score_bound=bound, **kw)
        e_run = close(o_run.float(), ref, 1.2e-2, f'running-max attention vs f64 softmax ({what})')
        e_fix = close(o_fix.float(), ref, 1.2e-2, f'fixed-offset attention vs f64 softmax ({what})')
        close(o_fix.float(), o_run.float(), 1.2e-2, f'fixed-offset vs running-max ({what})')
        assert e_fix <= 1.5 * e_run + 1e-3, f'{what}: the fixed offset must not cost accuracy ({e_fix:.2e} vs {e_run:.2e})'
        # a generous bound (12 above the true maximum) only shifts exponents
        o_far = torch.full_like(o_full, float('nan'))
        L.attn_fwd(L.BF16, Qp, Kp, Vt, o_far, S, heads, n, n, 0, score_bound=bound + 12, **kw)
        close(o_far.float(), ref, 1.2e-2, f'fixed-offset attention, loose bound ({what})')


@pytest.mark.parametrize('dims,heads', [((4, 9, 9), 2), ((3, 5, 7), 8), ((2, 6, 6), 1)])
def test_attention_fixed_offset_ignores_poisoned_pad_rows(L, dims, heads):
    """with n > 64 and n % 32 != 0 the K^ / V^T images have pad rows pk_qkv_project never writes (torch.empty:
    recycled memory, often NaN / Inf).  The fixed-offset softmax masked them through arithmetic (fma(NaN, c, -inf) = NaN) and a
    whole launch went NaN; padding must be masked by selection.  Pads are poisoned here with NaN, +Inf and -Inf."""
    from phenaki_pytorch_amd.attention import ContinuousPositionBias
    n = dims[0] * dims[1] * dims[2]
    assert n > 64 and n % 32 != 0
    S = 2
    torch.manual_seed(11)
    cpb = ContinuousPositionBias(dim=64, heads=heads, num_dims=3).cuda()
    nq_pad, nk_pad = L.attn_pads(n, n, 0)
    Qp = (torch.randn(S * heads, nq_pad, 64, generator=g(191)) * 0.35).cuda().to(torch.bfloat16)
    Kp = (torch.randn(S * heads, nk_pad, 64, generator=g(192)) * 0.35).cuda().to(torch.bfloat16)
    Vt = torch.randn(S * heads, 64, nk_pad, generator=g(193)).cuda().to(torch.bfloat16)
    Q = Qp[:, :n].double().cpu()
    K = Kp[:, :n].double().cpu()
    V = Vt[:, :, :n].double().cpu()
    for tag, kw, extra in (('no bias', {}, 0.), ('table', dict(bias_table=cpb.table(*dims)), cpb(*dims).double().cpu().repeat(S, 1, 1))):
        sim = Q @ K.transpose(1, 2) + extra
        ref = (sim.softmax(-1) @ V.transpose(1, 2)).view(S, heads, n, 64).permute(0, 2, 1, 3).reshape(S * n, heads * 64)
        bound = float(sim.max()) + 0.3
        for poison in (float('nan'), float('inf'), float('-inf')):
            Kq, Vq = Kp.clone(), Vt.clone()
            Kq[:, n:] = poison
            Vq[:, :, n:] = poison
            Qq = Qp.clone()
            Qq[:, n:] = poison                                   # pad QUERY rows: computed, never stored
            for sb in (bound, None):                             # fixed-offset and running-max loops
                o = torch.full((S * n, heads * 64), float('nan'), device='cuda', dtype=torch.bfloat16)
                L.attn_fwd(L.BF16, Qq.reshape(-1), Kq.reshape(-1), Vq.reshape(-1), o, S, heads, n, n, 0, score_bound=sb, **kw)
                close(o.float(), ref, 1.2e-2, f'{tag}, pads = {poison}, {"fixed offset" if sb else "running max"}')


@pytest.mark.parametrize('dims,S,heads', [((9, 8, 8), 2, 8), ((4, 9, 9), 2, 2), ((3, 8, 8), 3, 2), ((2, 9, 12), 1, 1)])
def test_attention_split_bf16_fixed_offset_lds_kernel(L, dims, S, heads):
    """round 3: the LDS-staged fixed-offset attention on split-bf16 images (pk_attn_fwd dtype 2 with score_bound, 128-row workgroups, bias
    table in LDS or no bias) against an f64 softmax of the same f32 operands -- f32-grade -- and against the LDS-free running-max kernel;
    pad rows / columns poisoned (n % 64 != 0 cases)."""
    from phenaki_pytorch_amd.attention import ContinuousPositionBias
    n = dims[0] * dims[1] * dims[2]
    torch.manual_seed(12)
    cpb = ContinuousPositionBias(dim=64, heads=heads, num_dims=3).cuda()
    nq_pad, nk_pad = L.attn_pads(n, n, 0)
    Q32 = torch.randn(S * heads, nq_pad, 64, generator=g(291)) * 0.35
    K32 = torch.randn(S * heads, nk_pad, 64, generator=g(292)) * 0.35
    V32 = torch.randn(S * heads, 64, nk_pad, generator=g(293))
    Q, K, V = Q32[:, :n].double(), K32[:, :n].double(), V32[:, :, :n].double()
    for poison in (0.0, float('nan'), float('inf')):
        Qx, Kx, Vx = Q32.clone(), K32.clone(), V32.clone()
        if n < nq_pad:
            Qx[:, n:] = poison
        if n < nk_pad:
            Kx[:, n:] = poison
            Vx[:, :, n:] = poison
        Qp = L.split_planes(Qx.reshape(-1, 64).cuda()).reshape(-1)
        Kp = L.split_planes(Kx.reshape(-1, 64).cuda()).reshape(-1)
        Vt = L.split_planes(Vx.reshape(-1, nk_pad).cuda()).reshape(-1)
        for tag, kw, extra in (('no bias', {}, 0.), ('table', dict(bias_table=cpb.table(*dims)), cpb(*dims).double().cpu().repeat(S, 1, 1))):
            sim = Q @ K.transpose(1, 2) + extra
            ref = (sim.softmax(-1) @ V.transpose(1, 2)).view(S, heads, n, 64).permute(0, 2, 1, 3).reshape(S * n, heads * 64)
            bound = float(sim.max()) + 0.3
            o_fix = torch.full((S * n, heads * 64), float('nan'), device='cuda')
            L.attn_fwd(L.BF16X3, Qp, Kp, Vt, o_fix, S, heads, n, n, 0, score_bound=bound, **kw)
            close(o_fix, ref, 2e-4, f'split-bf16 fixed-offset attention vs f64 softmax ({tag}, pads {poison}, {dims})')
            o_far = torch.full_like(o_fix, float('nan'))
            L.attn_fwd(L.BF16X3, Qp, Kp, Vt, o_far, S, heads, n, n, 0, score_bound=bound + 9, **kw)
            close(o_far, ref, 2e-4, f'split-bf16 fixed-offset attention, loose bound ({tag})')
            if tag == 'no bias':
                o_run = torch.full_like(o_fix, float('nan'))
                L.attn_fwd(L.BF16X3, Qp, Kp, Vt, o_run, S, heads, n, n, 0)
                close(o_fix, o_run, 2e-4, 'fixed-offset LDS kernel vs running-max kernel (split-bf16)')


@pytest.mark.parametrize('variant', [0, 3, 8, 9, 24, 27])
@pytest.mark.parametrize('M,N,K', [(300, 200, 96), (1000, 520, 1368), (4608, 512, 512), (129, 2736, 512), (512, 512, 6144), (77, 4, 64)])
def test_gemm_split_bf16(L, variant, M, N, K):
    """the split-bf16 ("bf16x3") main loops: f32 A rows split into (hi, lo) bf16 planes in registers, host-packed W planes, three bf16
    MFMAs per fragment pair.  Against an f64 product: per-product error ~2^-17, i.e. two orders below plain bf16 operands (whose error on
    the same data is measured beside it) and inside the f32 GEMM test's tolerance band; M / N / K tails, row gather, bias + residual,
    GEGLU and LeakyReLU epilogues, every tile variant."""
    A = torch.randn(M + 7, K, generator=g(160))
    W = torch.randn(N, K, generator=g(161)) / math.sqrt(K)
    bias = torch.randn(N, generator=g(162))
    res = torch.randn(M, N, generator=g(163))
    idx = torch.randperm(M + 7, generator=g(164))[:M].int()
    Kp = (K + 31) // 32 * 32
    Wp = torch.zeros(N, Kp)
    Wp[:, :K] = W
    Wd = L.split_planes(Wp.cuda())
    assert Wd.dtype == torch.float32 and tuple(Wd.shape) == (N, Kp)
    ref64 = A.double()[idx.long()] @ W.double().t()
    ref = (ref64 + bias.double() + res.double()).float()
    C = torch.full((M, N), float('nan'), device='cuda')
    L.gemm(L.BF16X3, A.cuda(), Wd, M, N, K, C=C, bias=bias.cuda(), res=res.cuda(), a_rows=idx.cuda(), variant=variant)
    e_split = close(C, ref, 4e-5, f'split gemm variant {variant} {M}x{N}x{K}')
    e_bf16 = ((bf(A)[idx.long()] @ bf(W).t() + bias + res) - ref).abs().max().item() / ref.abs().max().item()
    assert e_split < e_bf16 / 30, f'split-bf16 error {e_split:.2e} is not two orders below plain bf16 ({e_bf16:.2e})'
    C2 = torch.full((M, N), float('nan'), device='cuda')
    L.gemm(L.BF16X3, A.cuda(), Wd, M, N, K, C=C2, variant=variant)
    close(C2, (A.double()[:M] @ W.double().t()).float(), 4e-5, f'split gemm variant {variant} plain')
    if N % 2 == 0 and N >= 8:
        h = A.double()[:M] @ W.double().t()
        C3 = torch.full((M, N // 2), float('nan'), device='cuda')
        L.gemm(L.BF16X3, A.cuda(), Wd, M, N, K, C=C3, act=L.ACT_GEGLU, variant=variant)
        close(C3, (F.gelu(h[:, 1::2]) * h[:, 0::2]).float(), 4e-5, f'split gemm variant {variant} geglu')
        C4 = torch.full((M, N), float('nan'), device='cuda')
        L.gemm(L.BF16X3, A.cuda(), Wd, M, N, K, C=C4, bias=bias.cuda(), act=L.ACT_LEAKY, variant=variant)
        close(C4, F.leaky_relu(h + bias.double(), 0.1).float(), 4e-5, f'split gemm variant {variant} leaky')
    # bf16-mode features do not exist for it, and A must be f32
    with pytest.raises(RuntimeError, match='PK_EINVAL'):
        L.gemm(L.BF16X3, A.cuda().to(torch.bfloat16), Wd, M, N, K, C=C2, variant=variant)


@pytest.mark.parametrize('S,h,n,nkv,nnull,causal', [(3, 2, 64, 64, 0, False), (2, 8, 200, 200, 0, False), (2, 2, 37, 13, 2, False), (5, 2, 9, 9, 0, True), (1, 8, 576, 576, 0, False)])
def test_attention_split_bf16_images(L, S, h, n, nkv, nnull, causal):
    """pk_attn_prep / pk_attn_fwd in split-bf16 mode: the operand images are pre-split (hi | lo) planes in 128-byte blocks; the result must
    sit at f32 level against an f64 softmax attention of the same q / k / v (and against the exact-f32 kernels)."""
    dim = h * 64
    q = torch.randn(S * n, dim, generator=g(170))
    kv = torch.randn(S * nkv, 2 * dim, generator=g(171))
    null_kv = torch.randn(h, 2 * max(nnull, 1), 64, generator=g(172))
    qs, ks = torch.rand(64, generator=g(173)) + 0.5, torch.rand(64, generator=g(174)) + 0.5
    bias = torch.randn(h, n, nkv, generator=g(175)) if not causal else None
    km = (torch.rand(S, nkv, generator=g(176)) > 0.25) if nnull else None
    slopes = (torch.rand(h, generator=g(177)) * 0.5).reshape(h, 1, 1) if causal else None
    outs = {}
    for name, dt in (('f32', L.F32), ('split', L.BF16X3)):
        nq_pad, nk_pad = L.attn_pads(n, nkv, nnull)
        Qp = torch.full((S * h * nq_pad * 64,), float('nan'), device='cuda')
        Kp = torch.full((S * h * nk_pad * 64,), float('nan'), device='cuda')
        Vt = torch.full((S * h * nk_pad * 64,), float('nan'), device='cuda')
        L.attn_prep(dt, q.cuda(), kv.cuda(), null_kv.cuda(), qs.cuda(), ks.cuda(), 8.0, Qp, Kp, Vt, S, h, n, nkv, nnull)
        O_ = torch.full((S * n, dim), float('nan'), device='cuda')
        L.attn_fwd(dt, Qp, Kp, Vt, O_, S, h, n, nkv, nnull, bias=bias.cuda() if bias is not None else None,
                   kmask=km.to(torch.uint8).cuda() if km is not None else None, slopes=slopes.cuda() if slopes is not None else None, causal=causal)
        outs[name] = O_.cpu()
    # f64 reference
    qd = F.normalize(q.double().view(S, n, h, 64).permute(0, 2, 1, 3), dim=-1) * qs.double() * 8.0
    kd = kv.double()[:, :dim].view(S, nkv, h, 64).permute(0, 2, 1, 3)
    vd = kv.double()[:, dim:].view(S, nkv, h, 64).permute(0, 2, 1, 3)
    if nnull:
        nk_, nv_ = null_kv.double()[:, 0::2][:, :nnull], null_kv.double()[:, 1::2][:, :nnull]
        kd = torch.cat((nk_[None].expand(S, -1, -1, -1), kd), dim=2)
        vd = torch.cat((nv_[None].expand(S, -1, -1, -1), vd), dim=2)
    kd = F.normalize(kd, dim=-1) * ks.double()
    sim = qd @ kd.transpose(-1, -2)
    if bias is not None:
        sim[..., nnull:] += bias.double()
    if km is not None:
        full = torch.cat((torch.ones(S, nnull, dtype=torch.bool), km), dim=1)
        sim = sim.masked_fill(~full[:, None, None, :], -torch.finfo(torch.float32).max)
    if causal:
        i, j = torch.arange(n)[:, None], torch.arange(nkv)[None, :]
        sim = sim - (j - i).abs().double() * slopes.double().view(1, h, 1, 1)
        sim = sim.masked_fill((j > i)[None, None], -torch.finfo(torch.float32).max)
    ref = (sim.softmax(-1) @ vd).permute(0, 2, 1, 3).reshape(S * n, dim).float()
    e32 = close(outs['f32'], ref, 2e-5, 'exact-f32 attention vs f64')
    e3 = close(outs['split'], ref, 1e-4, 'split-bf16 attention vs f64')
    print(f'attention n={n}: f32 err {e32:.2e}, split-bf16 err {e3:.2e}')


@pytest.mark.parametrize('B,C,Fr,H,W,pt,ph,pw,N', [(2, 3, 5, 64, 64, 2, 32, 32, 128), (1, 3, 3, 256, 256, 2, 32, 32, 512), (2, 3, 5, 64, 64, 2, 16, 16, 128),
                                                   (3, 3, 1, 64, 128, 2, 32, 64, 64), (2, 3, 3, 32, 32, 2, 16, 8, 72), (5, 3, 5, 96, 96, 2, 32, 32, 132)])
def test_patch_embed_fused(L, B, C, Fr, H, W, pt, ph, pw, N):
    """pk_patch_embed (K1): patch gather + LayerNorm(P) + Linear in one launch for both frame groups, against the reference op sequence
    (Rearrange -> LayerNorm(P) -> Linear, cvivit.py:273-285) in f64 -- including a video with a large DC offset and little contrast, where
    rounding the un-centred pixels to bf16 would lose the signal: the kernel centres every patch on its first element before rounding."""
    nh, nw = H // ph, W // pw
    nt = (Fr - 1) // pt
    for offset, contrast in ((0.0, 1.0), (40.0, 0.05)):
        video = torch.randn(B, C, Fr, H, W, generator=g(300)) * contrast + offset
        groups, refs, outs = [], [], []
        for f0, ntg, ptg, seed in ((1, nt, pt, 1), (0, 1, 1, 2)):
            if ntg <= 0:
                continue
            P = C * ptg * ph * pw
            gamma = 1 + 0.1 * torch.randn(P, generator=g(301 + seed))
            beta = 0.1 * torch.randn(P, generator=g(303 + seed))
            Wl = torch.randn(N, P, generator=g(305 + seed)) / math.sqrt(P)
            bl = 0.1 * torch.randn(N, generator=g(307 + seed))
            fr = video[:, :, f0:f0 + ntg * ptg]
            pat = fr.reshape(B, C, ntg, ptg, nh, ph, nw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * ntg * nh * nw, P).double()
            ref = F.layer_norm(pat, (P,), gamma.double(), beta.double()) @ Wl.double().t() + bl.double()
            Kp = (P + 63) // 64 * 64
            wg = torch.zeros(N, Kp)
            wg[:, :P] = Wl * gamma[None, :]
            wg = wg.to(torch.bfloat16)
            s = wg.float().sum(dim=1)
            t = Wl @ beta + bl
            out = torch.full((B * ntg * nh * nw, N), float('nan'), device='cuda')
            groups.append((wg.cuda(), s.cuda(), t.cuda(), out, f0, ntg, ptg))
            refs.append(ref.float())
            outs.append(out)
        L.patch_embed(video.cuda(), ph, pw, N, groups)
        for gi, (out, ref) in enumerate(zip(outs, refs)):
            close(out, ref, 1.2e-2, f'patch embed group {gi} offset {offset}')           # bf16 operands: ~2^-9 per product over P terms
            rms = ((out.cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            assert rms < 4e-3, f'group {gi} offset {offset}: rms error {rms:.2e}'
    # a single group (single-frame input) and argument checks
    video = torch.randn(B, C, 1, H, W, generator=g(310))
    P = C * ph * pw
    if P % 192 == 0:
        wg = (torch.randn(N, P, generator=g(311)) / math.sqrt(P)).to(torch.bfloat16)
        out = torch.full((B * nh * nw, N), float('nan'), device='cuda')
        L.patch_embed(video.cuda(), ph, pw, N, [(wg.cuda(), wg.float().sum(1).cuda(), torch.zeros(N).cuda(), out, 0, 1, 1)])
        pat = video.reshape(B, C, 1, 1, nh, ph, nw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * nh * nw, P).double()
        close(out, (F.layer_norm(pat, (P,)) @ wg.double().t()).float(), 1.2e-2, 'single group')
        with pytest.raises(RuntimeError, match='PK_EINVAL'):
            L.patch_embed(video.cuda(), ph, pw, N, [(wg.cuda(), wg.float().sum(1).cuda(), torch.zeros(N).cuda(), out, 0, 2, 1)])      # frames beyond F


@pytest.mark.parametrize('B,C,Fr,H,W,pt,ph,pw,N', [(2, 3, 5, 64, 64, 2, 16, 16, 128), (1, 3, 17, 256, 256, 2, 32, 32, 512), (3, 3, 3, 32, 64, 2, 8, 32, 96),
                                                   (2, 1, 1, 64, 64, 2, 32, 32, 64), (1, 3, 9, 96, 64, 2, 32, 32, 512)])
def test_patch_embed_row_panels_splitk(L, B, C, Fr, H, W, pt, ph, pw, N):
    """pk_patch_embed_splitk + pk_patch_embed_finish (round 4: row panels x all columns x K-slices, the LayerNorm(dim) that follows fused into the
    finish) against the reference op sequence Rearrange -> LayerNorm(P) -> Linear -> LayerNorm(dim) (cvivit.py:273-285) in f64, both frame groups,
    ragged row panels and short last slices, f32 and bf16 token rows with the (b t h w) remap, incl. the low-contrast / large-offset video."""
    nh, nw = H // ph, W // pw
    nt = (Fr - 1) // pt
    T = 1 + nt
    hw = nh * nw
    for offset, contrast in ((0.0, 1.0), (40.0, 0.05)):
        video = torch.randn(B, C, Fr, H, W, generator=g(300)) * contrast + offset
        tokens = torch.full((B * T * hw, N), float('nan'), device='cuda')
        tokens_t = torch.full((B * T * hw, N), float('nan'), device='cuda', dtype=torch.bfloat16)
        want = torch.zeros(B, T, hw, N, dtype=torch.float64)
        spec, fin = [], []
        for f0, ntg, ptg, seed, goff in ((1, nt, pt, 1, hw), (0, 1, 1, 2, 0)):
            if ntg <= 0:
                continue
            P = C * ptg * ph * pw
            gamma = 1 + 0.1 * torch.randn(P, generator=g(301 + seed))
            beta = 0.1 * torch.randn(P, generator=g(303 + seed))
            Wl = torch.randn(N, P, generator=g(305 + seed)) / math.sqrt(P)
            bl = 0.1 * torch.randn(N, generator=g(307 + seed))
            g2, b2 = 1 + 0.1 * torch.randn(N, generator=g(309 + seed)), 0.1 * torch.randn(N, generator=g(311 + seed))
            fr = video[:, :, f0:f0 + ntg * ptg]
            pat = fr.reshape(B, C, ntg, ptg, nh, ph, nw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * ntg * hw, P).double()
            ref = F.layer_norm(F.layer_norm(pat, (P,), gamma.double(), beta.double()) @ Wl.double().t() + bl.double(), (N,), g2.double(), b2.double())
            want[:, (0 if f0 == 0 else 1):(1 if f0 == 0 else T)] = ref.view(B, ntg, hw, N)
            Kp = (P + 63) // 64 * 64
            wg = torch.zeros(N, Kp)
            wg[:, :P] = Wl * gamma[None, :]
            wg = wg.to(torch.bfloat16)
            ns, rows = L.patch_embed_slices(P), B * ntg * hw
            assert ns == (P + 1023) // 1024
            part = torch.full((ns, rows, N), float('nan'), device='cuda')
            stats = torch.full((ns, rows, 2), float('nan'), device='cuda')
            spec.append((wg.cuda(), part, stats, f0, ntg, ptg))
            fin.append((part, stats, P, wg.float().sum(1).cuda(), (Wl @ beta + bl).cuda(), g2.cuda(), b2.cuda(), (ntg * hw, T * hw, goff)))
        L.patch_embed_splitk(video.cuda(), ph, pw, N, spec)
        for part, stats, P, s, t, g2, b2, remap in fin:
            assert torch.isfinite(part).all() and torch.isfinite(stats).all()
            L.patch_embed_finish(part, stats, P, s, t, 1e-5, g2, b2, 1e-5, out2=tokens, out=tokens_t, remap=remap)
        ref = want.view(B * T * hw, N).float()
        close(tokens, ref, 2e-2, f'tokens, offset {offset}')           # bf16 operands over P terms, then a LayerNorm
        rms = ((tokens.cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        assert rms < 5e-3, f'offset {offset}: rms error {rms:.2e}'
        assert torch.equal(tokens_t.float().cpu(), bf(tokens.cpu())), 'the bf16 token rows are the rounding of the f32 ones'
        # round 5: both groups folded by ONE launch (pk_patch_embed_finish_groups): bit-identical token rows
        tok1 = torch.full_like(tokens, float('nan'))
        tok1_t = torch.zeros_like(tokens_t)
        L.patch_embed_finish_groups([(part, stats, P, s, t, 1e-5, g2, b2, 1e-5, remap) for part, stats, P, s, t, g2, b2, remap in fin], out2=tok1, out=tok1_t)
        assert torch.equal(tok1, tokens) and torch.equal(tok1_t, tokens_t)


@pytest.mark.parametrize('numel', [1, 7, 1000, 70000, 256 * 2048 * 4 + 5, 3 * 576 * 65536])
def test_torch_philox_reproduction_is_bit_exact(L, numel):
    """common.hpp torch_uniform == torch.zeros(numel, device='cuda').uniform_(0, 1), element by element, through pk_vocab_sample_philox:
    with W = 0 and bias = 0 every logit is 0, so pred[row] = argmax_v gumbel(u[row][v]) = argmax_v u[row][v] -- compared with the argmax of
    the materialised torch fill (all rows, and a gathered subset), at a non-zero generator offset; the generator ends where torch's op ends"""
    V = 64 if numel < 100000 else (4096 if numel < 10 ** 7 else 65536)
    rows_total = max(1, numel // V)
    n_el = rows_total * V
    D = 32
    A = torch.zeros((rows_total, D), device='cuda')
    W = torch.zeros((V, D), device='cuda')
    bias = torch.zeros((V,), device='cuda')
    gen = torch.cuda.default_generators[0]
    torch.manual_seed(4242)
    torch.zeros(1000, device='cuda').uniform_(0, 1)                # move the generator off offset 0
    state = gen.get_state()
    ref = torch.zeros((rows_total, V), device='cuda').uniform_(0, 1)
    end = gen.get_offset()
    gen.set_state(state)
    spec = L.TorchPhilox(torch.device('cuda', 0), n_el)
    partials = torch.empty((5 * L.vocab_ntiles(V) * rows_total,), device='cuda')
    L.vocab_sample_philox(L.F32, A, W, bias, rows_total, V, D, 1.0, None, spec, False, partials)
    spec.advance()
    assert gen.get_offset() == end
    pred = torch.empty((rows_total,), device='cuda', dtype=torch.long)
    L.vocab_reduce(partials, rows_total, V, None, None, None, pred, None, False)
    want = ref.argmax(dim=-1)
    assert torch.equal(pred, want), f'{int((pred != want).sum())} of {rows_total} rows differ'
    if rows_total >= 8:
        rows = torch.tensor([rows_total - 1, 0, rows_total // 2, 3], dtype=torch.int32, device='cuda')
        Ag = torch.zeros((4, D), device='cuda')
        p2 = torch.empty((5 * L.vocab_ntiles(V) * 4,), device='cuda')
        L.vocab_sample_philox(L.F32, Ag, W, bias, 4, V, D, 1.0, rows, spec, False, p2)
        pr = torch.full((rows_total,), -1, device='cuda', dtype=torch.long)
        L.vocab_reduce(p2, 4, V, rows, None, None, pr, None, False)
        assert torch.equal(pr[rows.long()], want[rows.long()])


def test_torch_library_ops_match_the_direct_wrappers(L):
    """phenaki_mi355x::gemm / layernorm / vocab_sample + vocab_reduce through the dispatcher == the ctypes wrappers the modules call"""
    import phenaki_pytorch_amd.ops  # noqa: F401
    from phenaki_pytorch_amd.attention import pack_linear_weight
    ops = torch.ops.phenaki_mi355x
    x = torch.randn(300, 512, generator=g(401)).cuda()
    w = (torch.randn(256, 512, generator=g(402)) / 22).cuda()
    b = torch.randn(256, generator=g(403)).cuda()
    for dt in (L.F32, L.BF16X3):
        wp = pack_linear_weight(w, dt)
        want = torch.empty((300, 256), device='cuda')
        L.gemm(dt, x, wp, 300, 256, 512, C=want, bias=b)
        got = ops.gemm(x, wp, b, None, torch.empty((300, 256), device='cuda'), dt, 0)
        assert torch.equal(got, want)
    gam = (torch.rand(512, generator=g(404)) + 0.5).cuda()
    y = ops.layernorm(x, gam, None, torch.empty_like(x), 1e-5)
    close(y.cpu(), torch.nn.functional.layer_norm(x.cpu(), (512,), gam.cpu(), None), 1e-5, 'layernorm op')
    V = 1024
    wv = pack_linear_weight((torch.randn(V, 512, generator=g(405)) / 22).cuda(), L.F32)
    bv = torch.zeros(V, device='cuda')
    p1 = torch.empty((5 * L.vocab_ntiles(V) * 300,), device='cuda')
    p2 = torch.empty_like(p1)
    L.vocab_sample(L.F32, x, wv, bv, 300, V, 512, 0.7, None, None, 1234, False, p1)
    ops.vocab_sample(x, wv, bv, None, None, p2, L.F32, 0.7, 1234, False)
    a1, a2 = torch.empty(300, device='cuda', dtype=torch.long), torch.empty(300, device='cuda', dtype=torch.long)
    L.vocab_reduce(p1, 300, V, None, None, None, a1, None, False)
    ops.vocab_reduce(p2, None, None, None, a2, None, V, False)
    assert torch.equal(a1, a2)
