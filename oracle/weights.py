"""Deterministic, name-keyed synthetic weights and inputs.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Full-size checkpoints (60 M + 95 M + 61 M parameters) cannot be committed as
fixtures and there is no network for real ones, so every party -- the golden
generator driving the REAL reference, the oracle, the HIP path under test and
bench.py -- fills a module's ``state_dict`` with the same values by hashing the
parameter NAME into a CPU generator seed.  torch's CPU generator is
bit-reproducible across machines for one torch build (same image here and on
the GPU box), so only outputs need to be stored under tests/golden/.
"""
import zlib
import torch


def _gen(name, salt=0):
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return g


def fill_value(name, ref, salt=0):
    """value for one state_dict entry, chosen by the entry's role (scales follow the
    reference's default initialisers so activations stay O(1))."""
    shape, g = tuple(ref.shape), _gen(name, salt)
    leaf = name.split('.')[-1]
    if not ref.dtype.is_floating_point:
        return None  # integer buffers (LFQ mask) keep their constructed value
    if ref.numel() == 0:
        return torch.zeros(shape)
    if leaf == 'beta' and 'norm' in name:       # attention.py:33 zero buffer -- keep it zero
        return torch.zeros(shape)
    if leaf in ('gamma',):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf in ('q_scale', 'k_scale'):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == 'null_kv':
        return torch.randn(shape, generator=g)
    if 'token_emb' in name or 'pos_emb' in name:
        return torch.randn(shape, generator=g)
    if leaf == 'weight' and len(shape) == 1:     # nn.LayerNorm weight
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == 'bias' and len(shape) == 1:
        return 0.05 * torch.randn(shape, generator=g)
    if leaf == 'weight' and len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        bound = (1.0 / fan_in) ** 0.5
        return (torch.rand(shape, generator=g) * 2 - 1) * bound * 1.7320508
    return 0.1 * torch.randn(shape, generator=g)


@torch.no_grad()
def fill_module(module, salt=0):
    """overwrite every floating-point parameter / persistent buffer of ``module`` in place."""
    sd = module.state_dict()
    for name, ref in sd.items():
        v = fill_value(name, ref, salt)
        if v is not None:
            ref.copy_(v.to(ref.dtype))
    return module


def synthetic_video(batch, frames, height, width, channels=3, seed=0):
    g = torch.Generator(device='cpu')
    g.manual_seed(1000 + seed)
    return torch.randn(batch, channels, frames, height, width, generator=g)


def synthetic_context(batch, length, dim, seed=1, pad_last=0):
    """cached 'T5' context (phenaki_pytorch/t5.py:64-103 boundary): rows of zeros are pads."""
    g = torch.Generator(device='cpu')
    g.manual_seed(2000 + seed)
    ctx = torch.randn(batch, length, dim, generator=g)
    if pad_last:
        ctx[-1, length - pad_last:] = 0.
    return ctx


def uniform_noise(shape, seed):
    g = torch.Generator(device='cpu')
    g.manual_seed(3000 + seed)
    return torch.rand(shape, generator=g)


def ragged_context(lengths, dim, seed=1):
    """what t5_encode_text hands over for captions of different lengths (t5.py:94-103): (B, max(lengths), dim) with row b's positions
    >= lengths[b] zero-filled -- Phenaki derives the per-row text_mask from exactly those zeros (phenaki_pytorch.py:461)."""
    ctx = synthetic_context(len(lengths), max(lengths), dim, seed=seed)
    for b, n in enumerate(lengths):
        ctx[b, n:] = 0.
    return ctx


def stub_vgg(image_size, salt=9):
    """a small stand-in for the perceptual network a caller passes as CViViT(vgg=...) (cvivit.py:346-347): (B, 3, H, W) -> (B, 32) features
    through pooling and two Linears (no convolution: the GPU box runs it on ATen without a MIOpen find step); name-keyed weights"""
    from torch import nn
    H, W = (image_size, image_size) if isinstance(image_size, int) else image_size
    net = nn.Sequential(nn.AvgPool2d(4), nn.Flatten(), nn.Linear(3 * (H // 4) * (W // 4), 64), nn.Tanh(), nn.Linear(64, 32))
    fill_module(net, salt=salt)
    for p_ in net.parameters():
        p_.requires_grad_(False)
    return net
