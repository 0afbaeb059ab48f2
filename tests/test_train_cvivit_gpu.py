"""The tokenizer's training step (phenaki_pytorch_amd/train_cvivit.py) against the REAL reference's autograd (tiny golden,
oracle/make_golden.py cvivit_grads_golden) and, at BASELINE geometry, against torch autograd through the reference-pinned oracle."""
import pytest
import torch

from oracle import phenaki_oracle as O
from oracle import weights
from oracle.configs import FULL, TINY, oracle_cfgs, state_dicts
from tests.test_oracle_golden import cvivit_grad_check, load
from tests.util import close, load_product, record_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    torch.cuda.set_device(0)
    with torch.enable_grad():
        yield


@pytest.mark.parametrize('kind', ['video', 'image', 'masked'])
@pytest.mark.parametrize('dtype,tol', [('fp32', 1e-3), ('bf16x3', 1e-3), ('bf16', None)])
def test_cvivit_training_step_matches_reference_autograd(golden_dir, dtype, tol, kind):
    """loss = cvivit(video); loss.backward() on the MI355X kernels == the reference's CViViT(use_vgg_and_gan=False).train() step: the loss and
    every parameter gradient (norm + strided sample).  bf16: the codes of near-zero projections may flip, so only the loss (2 %) and finite
    gradients of the same parameter set are required."""
    g = load(golden_dir, 'cvivit_grads_tiny.pt')
    cv, _, _, _ = load_product('tiny', TINY, dtype=dtype)
    cv.train()
    H = TINY['cvivit']['image_size']
    video = weights.synthetic_video(2, 5, H, H, seed=8).cuda()
    x = video[:, :, 2] if kind == 'image' else video
    loss = cv(x, mask=g['mask'].cuda()) if kind == 'masked' else cv(x)     # masked: variable-length training, the loss over the kept frames
    assert loss.requires_grad and loss.ndim == 0
    loss.backward()
    ref_loss = float(g[f'loss_{kind}'])
    named = dict(cv.named_parameters())
    grads = {k: v for k, v in g[f'grads_{kind}'].items() if v['norm'] > 0}
    if tol is None:
        assert abs(float(loss.detach()) - ref_loss) <= 2e-2 * ref_loss
        for k in grads:
            if named[k].numel():
                assert named[k].grad is not None and torch.isfinite(named[k].grad).all(), k
        return
    assert abs(float(loss.detach()) - ref_loss) <= 1e-4 * ref_loss
    cvivit_grad_check(lambda k: named[k].grad, grads, tol, 90 if kind == 'image' else 100)
    record_parity('cvivit_training_step_vs_reference_autograd', dict(dtype=dtype, kind=kind, loss=float(loss.detach()), ref_loss=ref_loss,
                                                                     parameters=len(grads)))


def test_cvivit_forward_routes_and_returns_recons(golden_dir):
    """grad mode + trainable parameters -> the training step; no_grad / frozen -> the plain value; return_recons gives the reconstruction"""
    cv, _, _, _ = load_product('tiny', TINY)
    H = TINY['cvivit']['image_size']
    video = weights.synthetic_video(2, 5, H, H, seed=8).cuda()
    cv.train()
    loss, recon = cv(video, return_recons=True)
    assert loss.requires_grad and recon.shape == video.shape and not recon.requires_grad
    with torch.no_grad():
        value, recon_v = cv(video, return_recons=True)
    assert not value.requires_grad
    assert abs(float(value) - float(loss.detach())) <= 1e-5 * float(value)
    close(recon, recon_v, 1e-4, 'reconstruction of the training forward vs the inference forward')
    assert abs(float(((recon - video) ** 2).mean()) - float(loss.detach())) <= 1e-5 * float(value)
    full = cv(video, mask=torch.ones(2, 5, dtype=torch.bool, device='cuda'))            # an all-true frame mask is the plain loss
    assert abs(float(full.detach()) - float(loss.detach())) <= 1e-6 * float(value)
    ids = cv(video, return_only_codebook_ids=True)                       # the inference surface is untouched by grad mode
    assert ids.dtype == torch.int64 and not ids.requires_grad
    for p in cv.parameters():
        p.requires_grad_(False)
    assert not cv(video).requires_grad


@pytest.mark.parametrize('dtype,tol', [('fp32', 1e-3), ('bf16x3', 1e-3)])
def test_cvivit_training_step_full_config_matches_oracle_autograd(dtype, tol):
    """BASELINE geometry (256 x 256, patch 32 x 32 x 2, dim 512, depth 4 + 4 twice, 8 heads, 17 frames -> 576 tokens, 16-bit LFQ): every
    gradient against torch autograd through the reference-pinned oracle on CPU"""
    cv_sd, _, _ = state_dicts('full')
    cvc, _, _ = oracle_cfgs(FULL)
    cv, _, _, _ = load_product('full', FULL, dtype=dtype)
    cv.train()
    video = weights.synthetic_video(1, 17, 256, 256, seed=9)
    leaf = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith('.beta') else v) for k, v in cv_sd.items()}
    ref = O.cvivit_recon_loss_train(leaf, cvc, video)
    ref.backward()
    loss = cv(video.cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-4 * float(ref.detach())
    top = max(float(v.grad.abs().max()) for v in leaf.values() if getattr(v, 'grad', None) is not None and v.numel())
    errs = {}
    for name, prm in cv.named_parameters():
        r = leaf[name]
        if r.grad is None or r.numel() == 0:
            continue
        assert prm.grad is not None, name
        if float(r.grad.abs().max()) < 1e-6 * top:
            assert float(prm.grad.abs().max()) <= 1e-2 * tol * top, name
            continue
        errs[name] = close(prm.grad.cpu(), r.grad, tol, f'd {name} ({dtype})')
    assert len(errs) >= 200
    worst = max(errs, key=errs.get)
    record_parity('cvivit_training_step_full_vs_oracle_autograd', dict(dtype=dtype, parameters=len(errs), worst=worst, worst_rel_err=errs[worst],
                                                                       median_rel_err=sorted(errs.values())[len(errs) // 2]))


def test_cvivit_training_loop_reduces_the_reconstruction_loss():
    """a few HipAdamW steps on one batch: the loss goes down (the step trains what it differentiates)"""
    import phenaki_pytorch_amd as P
    cv, _, _, _ = load_product('tiny', TINY, dtype='bf16x3')
    cv.train()
    H = TINY['cvivit']['image_size']
    video = weights.synthetic_video(2, 5, H, H, seed=8).cuda()
    opt = P.get_optimizer(cv.parameters(), lr=3e-4, wd=0.)
    losses = []
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        loss = cv(video)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(l == l for l in losses) and losses[-1] < 0.9 * losses[0], losses


@pytest.mark.parametrize('geom', ['non_square', 'temporal_patch_3', 'single_video_long'])
def test_cvivit_training_step_other_geometries_match_oracle_autograd(geom):
    """the row maps of the step ('(b t)(h w)' <-> '(b h w) t', first-frame | rest) and the patch-layout loss away from the square 4 x 4 / 8 x 8
    grids: h != w with different patch extents, temporal patch 3, one long video -- every gradient against autograd through the oracle"""
    import phenaki_pytorch_amd as P
    cfg = dict(dim=128, codebook_size=256, image_size=(32, 96), patch_size=(8, 16), temporal_patch_size=2, spatial_depth=1, temporal_depth=1,
               dim_head=64, heads=2)
    batch, frames = 2, 5
    if geom == 'temporal_patch_3':
        cfg.update(image_size=(32, 32), patch_size=(8, 8), temporal_patch_size=3)
        frames = 7
    elif geom == 'single_video_long':
        cfg.update(image_size=(32, 32), patch_size=(16, 16))
        batch, frames = 1, 23
    cv = P.CViViT(use_vgg_and_gan=False, **cfg)
    weights.fill_module(cv, salt=7)
    sd = {k: v.detach().clone() for k, v in cv.state_dict().items()}
    cvc = dict(image_size=tuple(cfg['image_size']), patch_size=tuple(cfg['patch_size']), temporal_patch_size=cfg['temporal_patch_size'],
               spatial_depth=1, temporal_depth=1, heads=2, channels=3)
    H, W = cfg['image_size']
    video = weights.synthetic_video(batch, frames, H, W, seed=11)
    leaf = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith('.beta') else v) for k, v in sd.items()}
    ref = O.cvivit_recon_loss_train(leaf, cvc, video)
    ref.backward()
    cv = cv.cuda().train()
    P.set_compute_dtype(cv, 'fp32')
    loss = cv(video.cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-4 * float(ref.detach())
    top = max(float(v.grad.abs().max()) for v in leaf.values() if getattr(v, 'grad', None) is not None and v.numel())
    checked = 0
    for name, prm in cv.named_parameters():
        r = leaf[name]
        if r.grad is None or r.numel() == 0:
            continue
        assert prm.grad is not None, name
        if float(r.grad.abs().max()) < 1e-6 * top:
            continue
        close(prm.grad.cpu(), r.grad, 1e-3, f'd {name} ({geom})')
        checked += 1
    assert checked >= 50
