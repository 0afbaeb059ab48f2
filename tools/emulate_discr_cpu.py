"""Developer check (NOT a test, NOT a product path): runs the Discriminator's autograd graph (phenaki_pytorch_amd/discriminator.py) on the CPU
with the C-ABI calls it makes replaced by torch expressions, and compares logits, hinge + gradient-penalty loss and every
gradient with oracle/gan_oracle.py.  It validates the graph composition (weight re-ordering, transposition flags of _MM, the second-order
closure) before GPU minutes are spent on the kernels themselves; the kernels are tested in tests/test_gan_gpu.py."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L                     # noqa: E402


def gemm(dtype, A, W, M, N, K, *, C, bias=None, res=None, act=0, **kw):
    y = A[:M, :K] @ W[:N, :K].t()
    if bias is not None:
        y = y + bias
    if act == L.ACT_LEAKY:
        y = F.leaky_relu(y, 0.1)
    if res is not None:
        y = y + res
    C.copy_(y)
    return C


def pack(src, R, K, transpose, out, Kp, kind, rows=None):
    assert kind == 0 and rows is None
    out.zero_()
    out[:R, :K] = src.t()[:R, :K] if transpose else src[:R, :K]
    return out


def pack_job(src, R, K, transpose, out, Kp, kind, **kw):
    assert not kw
    return (src, R, K, transpose, out, Kp, kind)


def pack_multi(jobs, like):
    for j in jobs:
        pack(*j)


def gemm_splitk(dtype, A, W, M, N, K, splits, C, bias=None, tile=0):
    Kc = K // splits
    for s in range(splits):
        C[s].copy_((A[:M, s * Kc:(s + 1) * Kc] @ W[:N, s * Kc:(s + 1) * Kc].t()).reshape(-1))
    return C


def sum_batch(src, S, out, E):
    out.copy_(src[:S].sum(0).reshape(out.shape))
    return out


def colsum(src, M, N, out, **kw):
    out.copy_(src[:M, :N].sum(0))
    return out


def leaky_bwd(y, dy, dz, M, N, slope=0.1):
    dz.copy_(dy * torch.where(y > 0, 1.0, slope))
    return dz


def im2col(x, B, H, W, C, kh, kw, stride, pad, cols):
    img = x.reshape(B, H, W, C).permute(0, 3, 1, 2)
    u = F.unfold(img, (kh, kw), padding=pad, stride=stride)               # (B, C kh kw, L) with row order (c, ky, kx)
    Lo = u.shape[-1]
    u = u.reshape(B, C, kh * kw, Lo).permute(0, 3, 2, 1).reshape(B * Lo, kh * kw * C)
    cols.copy_(u)
    return cols


def col2im(cols, B, H, W, C, kh, kw, stride, pad, dx):
    Lo = cols.shape[0] // B
    u = cols.reshape(B, Lo, kh * kw, C).permute(0, 3, 2, 1).reshape(B, C * kh * kw, Lo)
    img = F.fold(u, (H, W), (kh, kw), padding=pad, stride=stride)
    dx.copy_(img.permute(0, 2, 3, 1).reshape(B * H * W, C))
    return dx


def nchw_to_rows(img, Cp, rows):
    B, C, H, W = img.shape
    rows.zero_()
    rows[:, :C] = img.permute(0, 2, 3, 1).reshape(-1, C)
    return rows


def rows_to_nchw(rows, Cp, img):
    B, C, H, W = img.shape
    img.copy_(rows[:, :C].reshape(B, H, W, C).permute(0, 3, 1, 2))
    return img


def bmm(A, B, C, tA, tB, batch, M, N, K, *, lda, ldb, ldc, sA=0, sB=0, sC=0, accumulate=False):
    a = A.transpose(-1, -2) if tA else A
    b = B.transpose(-1, -2) if tB else B
    C.copy_((a @ b).reshape(C.shape))
    return C


def row_softmax(a, b, c, out, mode):
    if mode == 0:
        out.copy_(a.softmax(-1))
    elif mode == 1:
        out.copy_(a * (b - (a * b).sum(-1, keepdim=True)))
    else:
        out.copy_(c * (b - (a * b).sum(-1, keepdim=True)) - b * (c * a).sum(-1, keepdim=True))
    return out


def row_l2scale(x, sc, dz, gx, gsc, o0, o1, o2, mode):
    n = x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    u = x / n
    if mode == 0:
        o0.copy_(u * sc)
        return
    t = dz * sc
    a = (u * t).sum(-1, keepdim=True)
    if mode == 1:
        o0.copy_((t - u * a) / n)
        o1.copy_(dz * u)
        return
    P = lambda v: v - u * (u * v).sum(-1, keepdim=True)               # noqa: E731
    b = (gx * u).sum(-1, keepdim=True)
    pg = P(gx)
    o2.copy_(sc * pg / n + gsc * u)
    o1.copy_(dz * pg / n)
    o0.copy_(-(P(a * gx + b * t) + u * (gx * (t - u * a)).sum(-1, keepdim=True)) / n ** 2 + P(gsc * dz) / n)


def _ln_parts(x, eps):
    mu = x.mean(-1, keepdim=True)
    sig = (((x - mu) ** 2).mean(-1, keepdim=True) + eps).sqrt()
    return (x - mu) / sig, sig


def layernorm(x, gamma, beta, M, D, *, out=None, out2=None, raw=None, eps=1e-5, **kw):
    out2.copy_(F.layer_norm(x, (D,), gamma, beta, eps))


def layernorm_bwd(x, gamma, dy, dx, M, D, *, add=None, want_beta=False, eps=1e-5):
    xh, sig = _ln_parts(x, eps)
    gh = dy * gamma
    dx.copy_((gh - gh.mean(-1, keepdim=True) - xh * (gh * xh).mean(-1, keepdim=True)) / sig)
    return (dy * xh).sum(0), None


def row_ln_bwd2(x, gamma, dy, u, w, eps, grad_x, grad_gamma_rows, grad_dy):
    D = x.shape[-1]
    xh, sig = _ln_parts(x, eps)
    gh = dy * gamma
    c1, c2 = u.mean(-1, keepdim=True), (u * xh).mean(-1, keepdim=True)
    m1, m2 = gh.mean(-1, keepdim=True), (gh * xh).mean(-1, keepdim=True)
    gg = (u - c1 - xh * c2) / sig
    grad_dy.copy_(gamma * gg + w * xh)
    grad_gamma_rows.copy_(dy * gg)
    S = (u * gh).sum(-1, keepdim=True) - D * c1 * m1 - D * c2 * m2
    v = -(m2 * u + c2 * gh) / sig + w * dy
    grad_x.copy_((v - v.mean(-1, keepdim=True) - xh * (v * xh).mean(-1, keepdim=True)) / sig - S / sig ** 2 * xh / D)


EMULATED = dict(row_softmax=row_softmax, row_l2scale=row_l2scale, layernorm=layernorm, layernorm_bwd=layernorm_bwd, row_ln_bwd2=row_ln_bwd2,
                gemm=gemm, pack=pack, pack_job=pack_job, pack_multi=pack_multi, gemm_splitk=gemm_splitk, sum_batch=sum_batch, colsum=colsum, leaky_bwd=leaky_bwd, im2col=im2col,
                col2im=col2im, nchw_to_rows=nchw_to_rows, rows_to_nchw=rows_to_nchw, bmm=bmm, require_device=lambda t, name='tensor': None)


def install(setter=None):
    """replace the C-ABI wrappers the discriminator's graph calls by the torch expressions above (setter: e.g. pytest's monkeypatch.setattr)"""
    for name, fn in EMULATED.items():
        (setter or setattr)(L, name, fn)


def main(setter=None, cases=((32, 16), ((64, 32), 16), (64, 4))):
    install(setter)
    from oracle import gan_oracle as G
    from oracle import weights
    from phenaki_pytorch_amd.discriminator import Discriminator, gradient_penalty, hinge_discr_loss
    worst = 0.
    for size, dim in cases:
        torch.manual_seed(0)
        d = Discriminator(dim=dim, image_size=size)
        weights.fill_module(d, salt=1)
        H, W = (size, size) if isinstance(size, int) else size
        real = torch.randn(2, 3, H, W)
        fake = torch.randn(2, 3, H, W)
        # product graph
        rp = real.clone().requires_grad_()
        fl, rl = d(fake, second_order=True), d(rp, second_order=True)
        loss = hinge_discr_loss(fl, rl) + gradient_penalty(rp, rl)
        loss.backward()
        got = {k: v.grad.clone() for k, v in d.named_parameters() if v.grad is not None}
        # oracle
        sd = {'discr.' + k: v.detach().clone().requires_grad_(v.is_floating_point() and not k.endswith('beta')) for k, v in d.state_dict().items()}
        ro = real.clone().requires_grad_()
        flo, rlo = G.discriminator(sd, fake), G.discriminator(sd, ro)
        lo = G.hinge_discr_loss(flo, rlo) + G.gradient_penalty(ro, rlo)
        lo.backward()
        print(f'size {size} dim {dim}: logits {fl.detach().tolist()} vs {flo.detach().tolist()}; loss {float(loss.detach()):.6f} vs {float(lo.detach()):.6f}')
        assert torch.allclose(fl, flo, rtol=1e-4, atol=1e-5) and abs(float(loss) - float(lo)) <= 1e-4 * abs(float(lo))
        for k, gk in got.items():
            ref = sd['discr.' + k].grad
            if ref is None:
                assert float(gk.abs().max()) == 0, k
                continue
            err = float((gk - ref).abs().max()) / (float(ref.abs().max()) + 1e-20)
            worst = max(worst, err)
            assert err < 2e-3, (k, err)
        missing = [k for k, v in sd.items() if v.requires_grad and v.grad is not None and v.numel() and k[6:] not in got]
        assert not missing, missing
    print('ok: worst relative gradient error', worst)
    return worst


if __name__ == '__main__':
    main()
