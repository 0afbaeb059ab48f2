"""Drop the MI355X modules into the reference's OWN trainers, without editing the reference (north star: "so it drops into the existing trainers").

The reference's trainers type-check their model argument against the reference's classes (`@beartype class PhenakiTrainer: def __init__(self,
phenaki: Phenaki, ...)`, phenaki_pytorch/phenaki_trainer.py:174-178; `CViViTTrainer(vae: CViViT, ...)`, cvivit_trainer.py:56-60), and the annotation is
bound when the trainer module is imported.  `install()` therefore has to run BEFORE `import phenaki_pytorch`:

    import phenaki_pytorch_amd.dropin as dropin
    dropin.install()                                   # phenaki_pytorch.{CViViT, MaskGit, TokenCritic, SelfCritic, Phenaki, make_video} -> the HIP build
    from phenaki_pytorch import CViViT, Phenaki, CViViTTrainer, PhenakiTrainer      # the reference's trainers, annotated with / training the HIP modules

What it does: creates the `phenaki_pytorch` package object without running its `__init__` (which imports the trainers), imports the two leaf modules
that define the model classes (phenaki_pytorch/cvivit.py, phenaki_pytorch/phenaki_pytorch.py), rebinds the class names in them to this package's
classes, then executes the reference's `__init__` -- so `cvivit_trainer.py:21` (`from phenaki_pytorch.cvivit import CViViT`) and
`phenaki_trainer.py:35` (`from phenaki_pytorch.phenaki_pytorch import Phenaki`) resolve to the HIP classes and beartype's isinstance checks pass for real.

One more seam of the reference's control plane touches parameters behind autograd's back: `ema_pytorch.EMA.update` writes the averaged weights through
`.data` (no `_version` bump), which the packed-weight caches key on (attention.py: invalidate_packed).  When `ema_pytorch` is importable, `install()`
wraps `EMA.update` / `EMA.copy_params_from_model_to_ema` so the EMA copy's packed weights are dropped after every update
(cvivit_trainer.py:101-103, 262-263: `self.ema_vae = EMA(vae, ...)`, `self.ema_vae.update()`, then `ema_model(valid_data, return_recons_only=True)`).
"""
import importlib
import importlib.util
import sys
import types

_NAMES_CVIVIT = ('CViViT',)
_NAMES_PHENAKI = ('CViViT', 'MaskGit', 'TokenCritic', 'SelfCritic', 'Phenaki', 'make_video')


def installed(package='phenaki_pytorch'):
    m = sys.modules.get(package)
    return bool(m is not None and getattr(m, '_pk_dropin', False))


def _patch_ema():
    """ema_pytorch.EMA writes ema_model's parameters through `.data`: drop its packed-weight caches after every such write"""
    try:
        ema = importlib.import_module('ema_pytorch')
    except ImportError:
        return False
    from .attention import invalidate_packed
    cls = getattr(ema, 'EMA', None)
    if cls is None or getattr(cls, '_pk_dropin', False):
        return cls is not None
    for name in ('update', 'copy_params_from_model_to_ema', 'update_moving_average'):
        fn = getattr(cls, name, None)
        if fn is None:
            continue

        def wrapped(self, *a, _fn=fn, **kw):
            out = _fn(self, *a, **kw)
            target = getattr(self, 'ema_model', None)
            if target is not None:
                invalidate_packed(target)
            return out
        wrapped.__name__, wrapped.__doc__ = name, fn.__doc__
        setattr(cls, name, wrapped)
    cls._pk_dropin = True
    return True


def install(package='phenaki_pytorch'):
    """rebind the reference package's model classes to the MI355X build (see the module docstring); returns the package module.
    Raises if the reference package was already imported (its trainers' annotations are bound to the reference classes by then) or is absent."""
    if installed(package):
        return sys.modules[package]
    if package in sys.modules:
        raise RuntimeError(f'{package} is already imported: call phenaki_pytorch_amd.dropin.install() before the first `import {package}` '
                           '(the trainers bind their type annotations at import time)')
    spec = importlib.util.find_spec(package)
    if spec is None or not spec.submodule_search_locations:
        raise ImportError(f'the reference package `{package}` is not importable here')
    import phenaki_pytorch_amd as P
    pkg = types.ModuleType(package)
    pkg.__spec__, pkg.__path__, pkg.__file__, pkg.__package__ = spec, list(spec.submodule_search_locations), spec.origin, package
    pkg.__loader__ = spec.loader
    sys.modules[package] = pkg
    try:
        cv = importlib.import_module(package + '.cvivit')
        originals = {'cvivit': {n: getattr(cv, n) for n in _NAMES_CVIVIT}}
        for n in _NAMES_CVIVIT:
            setattr(cv, n, getattr(P, n))
        pp = importlib.import_module(package + '.phenaki_pytorch')
        originals['phenaki_pytorch'] = {n: getattr(pp, n) for n in _NAMES_PHENAKI if hasattr(pp, n)}
        for n in _NAMES_PHENAKI:
            setattr(pp, n, getattr(P, n))
        _patch_ema()
        spec.loader.exec_module(pkg)                  # the reference's own __init__: imports the trainers, now annotated with the HIP classes
    except BaseException:
        for k in [k for k in sys.modules if k == package or k.startswith(package + '.')]:
            del sys.modules[k]
        raise
    pkg._pk_dropin = True
    pkg._pk_reference_classes = originals              # the reference's own classes stay reachable (A/B runs, golden minting)
    return pkg
