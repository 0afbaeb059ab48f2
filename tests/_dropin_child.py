"""child process of tests/test_dropin_host.py: the reference's OWN trainers constructed around the MI355X modules (CPU, no kernel call).

Runs only where /root/reference exists (the build container).  Packages of the reference's control plane that this image lacks are stubbed
HERE (test infrastructure): `ema_pytorch` (same `.data.lerp_` update as the real package), `torchvision` (transforms / utils names only), `cv2`,
`vector_quantize_pytorch` (oracle/lfq.py) and a CHECKING `beartype` (isinstance on class-annotated arguments -- the property under test).
Prints one JSON line.
"""
import copy
import importlib.machinery
import json
import os
import sys
import tempfile
import types
import typing

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('PHENAKI_REFERENCE_ROOT', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import transformers  # noqa: E402,F401  (probes find_spec('torchvision') -- before the stub exists)

CHECKED = []          # (function, argument, annotated class, class of the value) of every isinstance check the beartype stub made


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _checking_beartype(obj):
    """isinstance checks on the arguments annotated with a plain class (what the real beartype enforces for `vae: CViViT`, `phenaki: Phenaki`)"""
    import functools
    import inspect
    if isinstance(obj, type):
        init = obj.__dict__.get('__init__')
        if init is not None:
            obj.__init__ = _checking_beartype(init)
        return obj
    fn = obj
    sig = inspect.signature(fn)

    @functools.wraps(fn)
    def inner(*a, **kw):
        try:
            hints = typing.get_type_hints(fn)
        except Exception:
            hints = {}
        bound = sig.bind(*a, **kw)
        for name, val in bound.arguments.items():
            h = hints.get(name)
            if isinstance(h, type) and issubclass(h, nn.Module):
                CHECKED.append((fn.__qualname__, name, h.__module__ + '.' + h.__qualname__, type(val).__module__ + '.' + type(val).__qualname__))
                if not isinstance(val, h):
                    raise TypeError(f'beartype: {fn.__qualname__}() parameter {name}={type(val).__module__}.{type(val).__qualname__} violates type hint {h}')
        return fn(*a, **kw)
    return inner


def _is_bearable(obj, hint):
    origin = typing.get_origin(hint)
    if origin is typing.Annotated:
        base, *preds = typing.get_args(hint)
        return isinstance(obj, base) and all(p(obj) for p in preds)
    if origin in (list, typing.List):
        (t,) = typing.get_args(hint)
        return isinstance(obj, list) and all(_is_bearable(o, t) for o in obj)
    if origin in (tuple, typing.Tuple):
        args = typing.get_args(hint)
        t = args[0]
        return isinstance(obj, tuple) and all(_is_bearable(o, t) for o in obj)
    return isinstance(obj, hint)


class _Is:
    def __class_getitem__(cls, fn):
        return fn


bt = _stub('beartype', beartype=_checking_beartype)
bt.door = _stub('beartype.door', is_bearable=_is_bearable)
bt.vale = _stub('beartype.vale', Is=_Is)


class _T:                               # torchvision.transforms stand-ins: identity, except Compose (chains) and ToTensor (PIL -> (C, H, W) in [0, 1])
    def __init__(self, *a, **kw):
        self.args = a

    def __call__(self, x):
        return x


class _Compose(_T):
    def __call__(self, x):
        for t in self.args[0]:
            x = t(x)
        return x


class _ToTensor(_T):
    def __call__(self, img):
        import numpy as np
        a = np.asarray(img.convert('RGB'), dtype=np.float32) / 255.
        return torch.from_numpy(a).permute(2, 0, 1).contiguous()


tv = _stub('torchvision')
tv.transforms = _stub('torchvision.transforms', Compose=_Compose, ToTensor=_ToTensor,
                      **{n: _T for n in ('Lambda', 'Resize', 'RandomHorizontalFlip', 'CenterCrop', 'ToPILImage')})
tv.datasets = _stub('torchvision.datasets', ImageFolder=_T)
tv.utils = _stub('torchvision.utils', make_grid=lambda *a, **k: None, save_image=lambda *a, **k: None)
_stub('cv2')
from oracle import lfq  # noqa: E402
_stub('vector_quantize_pytorch', LFQ=lfq.LFQ, VectorQuantize=lfq.VectorQuantize)


class EMA(nn.Module):
    """ema_pytorch.EMA reduced to what cvivit_trainer.py:101-103, 262-263 uses: a deep copy updated through `.data` (ema_pytorch's inplace_lerp /
    inplace_copy write `tgt.data`, which does NOT bump `_version`)"""

    def __init__(self, model, beta=0.9999, update_after_step=100, update_every=10, **kw):
        super().__init__()
        self.beta, self.online_model = beta, [model]
        self.ema_model = copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.register_buffer('step', torch.tensor(0))

    def update(self):
        self.step += 1
        for (_, ma), (_, cur) in zip(self.ema_model.named_parameters(), self.online_model[0].named_parameters()):
            ma.data.lerp_(cur.data, 1. - self.beta)


_stub('ema_pytorch', EMA=EMA)

# -------------------------------------------------------------------------------------------------------------------------------------------
import phenaki_pytorch_amd as P  # noqa: E402
from phenaki_pytorch_amd import dropin  # noqa: E402
from phenaki_pytorch_amd.attention import _cache  # noqa: E402

out = {}
pkg = dropin.install()
import phenaki_pytorch  # noqa: E402
from phenaki_pytorch import CViViT, CViViTTrainer, MaskGit, Phenaki, PhenakiTrainer, TokenCritic, make_video  # noqa: E402

out['names_are_hip_classes'] = bool(CViViT is P.CViViT and Phenaki is P.Phenaki and MaskGit is P.MaskGit and TokenCritic is P.TokenCritic
                                    and make_video is P.make_video and phenaki_pytorch is pkg)
out['trainers_are_the_reference_ones'] = bool(CViViTTrainer.__module__ == 'phenaki_pytorch.cvivit_trainer' and PhenakiTrainer.__module__ == 'phenaki_pytorch.phenaki_trainer'
                                              and sys.modules['phenaki_pytorch.cvivit_trainer'].__file__.startswith(REF))
out['reference_classes_kept'] = sorted(pkg._pk_reference_classes['phenaki_pytorch'])

tmp = tempfile.mkdtemp(prefix='pk_dropin_')
from phenaki_pytorch_amd.data import video_tensor_to_gif  # noqa: E402
os.makedirs(os.path.join(tmp, 'videos'))
for i in range(4):
    video_tensor_to_gif(torch.rand(3, 5, 32, 32, generator=torch.Generator().manual_seed(i)), os.path.join(tmp, 'videos', f'{i}.gif'))

cfg = dict(dim=64, codebook_size=256, image_size=32, patch_size=16, temporal_patch_size=2, spatial_depth=1, temporal_depth=1, dim_head=64, heads=2)
vae = CViViT(vgg=nn.Sequential(nn.Flatten(), nn.Linear(3 * 32 * 32, 8)), **cfg)           # GAN mode, as the reference's trainer expects (vae.discr)
trainer = CViViTTrainer(vae, folder=os.path.join(tmp, 'videos'), batch_size=2, num_train_steps=1, num_frames=5, results_folder=os.path.join(tmp, 'results_c'),
                        valid_frac=0.25, use_ema=True)
out['cvivit_trainer_constructed'] = True
out['cvivit_trainer_optimizers'] = [type(trainer.optim).__name__, type(trainer.discr_optim).__name__]
nv, nd = sum(p.numel() for p in trainer.vae_parameters), sum(p.numel() for p in vae.discr.parameters())
out['cvivit_param_split'] = [nv, nd, sum(p.numel() for p in vae.parameters())]
# EMA(vae): deepcopy of a PackedModule -- same keys, independent storage, the perceptual network shared state excluded from the checkpoint
ema = trainer.ema_vae
out['ema_deepcopy_ok'] = bool(set(ema.ema_model.state_dict()) == set(vae.state_dict()) and
                              all(a.data_ptr() != b.data_ptr() for a, b in zip(ema.ema_model.parameters(), vae.parameters()) if a.numel()))
lin = ema.ema_model.enc_spatial_transformer.layers[0][1].to_q
_cache(lin).get('probe', [lin.weight], lambda: lin.weight.detach().clone())              # a packed copy exists ...
had = '_pk_cache' in lin.__dict__
ema.update()                                                                            # ... the EMA writes lin.weight.data in place (no version bump) ...
out['ema_update_drops_packed_copies'] = bool(had and '_pk_cache' not in lin.__dict__)   # ... and dropin's wrapper dropped it
try:
    trainer.train_step()
    out['cvivit_first_step'] = 'ran (unexpected on a CPU-only host)'
except RuntimeError as e:
    out['cvivit_first_step'] = str(e)[:160]
# save() -> load() of the reference trainer round-trips through our state_dict (no vgg.* keys, discr.* kept)
ck = os.path.join(tmp, 'c.pt')
trainer.save(ck)
keys = set(torch.load(ck, weights_only=False)['model'])
out['checkpoint_keys_ok'] = bool(not any(k.startswith('vgg.') for k in keys) and any(k.startswith('discr.') for k in keys))
trainer.load(ck)

# a wrong class is still refused by the annotation (the check is real, not disabled)
try:
    CViViTTrainer(nn.Linear(2, 2), folder=os.path.join(tmp, 'videos'), batch_size=2, num_train_steps=1)
    out['wrong_class_refused'] = False
except TypeError:
    out['wrong_class_refused'] = True

mg = MaskGit(dim=64, num_tokens=256, max_seq_len=64, depth=1, heads=2, dim_head=64, unconditional=True)
cr = TokenCritic(dim=64, num_tokens=256, max_seq_len=64, depth=1, heads=2, dim_head=64, has_cross_attn=False)
ph = Phenaki(maskgit=mg, cvivit=CViViT(use_vgg_and_gan=False, **cfg), critic=cr, steps=2)
ptr = PhenakiTrainer(ph, folder=os.path.join(tmp, 'videos'), batch_size=2, num_frames=5, train_num_steps=1, results_folder=os.path.join(tmp, 'results_p'))
out['phenaki_trainer_constructed'] = True
out['phenaki_trainer_optimizer'] = type(ptr.opt).__name__
out['phenaki_opt_covers_maskgit'] = bool(sum(p.numel() for g in ptr.opt.param_groups for p in g['params']) == sum(p.numel() for p in mg.parameters()))
out['beartype_checks'] = CHECKED
print('DROPIN_JSON ' + json.dumps(out))
