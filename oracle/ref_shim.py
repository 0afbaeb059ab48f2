"""Import the REAL reference (read-only, /root/reference) without modifying it.

TEST INFRASTRUCTURE ONLY, and usable ONLY in the build container: the GPU box has
no /root/reference, so nothing under tests -m gpu / smoke / bench may call this.
It exists so that ``oracle/make_golden.py`` can mint golden vectors from the
reference itself:

  1. ``import transformers`` first (it probes find_spec('torchvision'));
  2. register a namespace module ``phenaki_pytorch`` whose __path__ is the reference
     package directory, so ``phenaki_pytorch/__init__.py`` (which drags in the trainers
     -> torchvision / ema_pytorch / cv2) is never executed;
  3. stub ``beartype`` (decorator only) and ``torchvision`` (VGG download only);
  4. provide ``vector_quantize_pytorch`` from ``oracle/lfq.py`` (the real package is absent);
  5. pre-seed the T5 config table so ``Phenaki.__init__`` does not hit the HF hub
     (phenaki_pytorch/phenaki_pytorch.py:391 evaluates get_encoded_dim eagerly).
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('PHENAKI_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'phenaki_pytorch'))


def load():
    """returns a namespace with the reference's CViViT, MaskGit, TokenCritic, Phenaki, make_video."""
    if not available():
        raise RuntimeError(f'reference not present at {REFERENCE_ROOT}')
    if 'phenaki_pytorch.phenaki_pytorch' in sys.modules and getattr(sys.modules['phenaki_pytorch'], '_oracle_shim', False):
        return _namespace()

    import transformers  # noqa: F401  (must precede the torchvision stub)

    if 'beartype' not in sys.modules:
        bt = types.ModuleType('beartype')
        bt.beartype = lambda f: f
        sys.modules['beartype'] = bt
    if 'torchvision' not in sys.modules:
        tv = types.ModuleType('torchvision')
        tv.__spec__ = importlib.machinery.ModuleSpec('torchvision', None)
        sys.modules['torchvision'] = tv
    if 'vector_quantize_pytorch' not in sys.modules:
        from oracle import lfq
        vq = types.ModuleType('vector_quantize_pytorch')
        vq.LFQ, vq.VectorQuantize = lfq.LFQ, lfq.VectorQuantize
        sys.modules['vector_quantize_pytorch'] = vq

    pkg = types.ModuleType('phenaki_pytorch')
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, 'phenaki_pytorch')]
    pkg._oracle_shim = True
    sys.modules['phenaki_pytorch'] = pkg

    t5 = importlib.import_module('phenaki_pytorch.t5')
    from transformers import T5Config
    t5.T5_CONFIGS[t5.DEFAULT_T5_NAME] = dict(config=T5Config(d_model=768))
    importlib.import_module('phenaki_pytorch.phenaki_pytorch')
    return _namespace()


def _namespace():
    m = sys.modules['phenaki_pytorch.phenaki_pytorch']
    ns = types.SimpleNamespace(
        CViViT=m.CViViT, MaskGit=m.MaskGit, TokenCritic=m.TokenCritic, Phenaki=m.Phenaki,
        make_video=m.make_video, module=m,
        attention=sys.modules['phenaki_pytorch.attention'], cvivit=sys.modules['phenaki_pytorch.cvivit'])
    return ns
