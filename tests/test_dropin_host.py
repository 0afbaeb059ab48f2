"""The reference's OWN trainers around the MI355X modules (north star "drops into the existing trainers").

`phenaki_pytorch_amd.dropin.install()` rebinds the reference package's model classes before its trainers are imported, so
`CViViTTrainer(vae: CViViT, ...)` (cvivit_trainer.py:56-60) and `PhenakiTrainer(phenaki: Phenaki, ...)` (phenaki_trainer.py:174-178) -- checked
by a beartype stand-in that really runs isinstance -- accept the HIP modules; constructed here on the CPU up to the first kernel call
(accelerate.prepare, the optimizer split over `vae.discr`, EMA(vae) = deepcopy of a PackedModule, save / load).  Needs /root/reference: CPU-only
container test, skipped on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('PHENAKI_REFERENCE_ROOT', '/root/reference')


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'phenaki_pytorch')), reason='the reference checkout is not present on this machine')
def test_reference_trainers_accept_the_hip_modules():
    env = dict(os.environ, CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_dropin_child.py')], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('DROPIN_JSON ')]
    assert r.returncode == 0 and line, f'child failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}'
    out = json.loads(line[-1][len('DROPIN_JSON '):])
    assert out['names_are_hip_classes'] and out['trainers_are_the_reference_ones']
    assert out['reference_classes_kept'] == ['CViViT', 'MaskGit', 'Phenaki', 'SelfCritic', 'TokenCritic', 'make_video']
    assert out['cvivit_trainer_constructed'] and out['phenaki_trainer_constructed']
    nv, nd, total = out['cvivit_param_split']
    assert nv > 0 and nd > 0 and nv + nd == total, 'set(vae.parameters()) - set(vae.discr.parameters()) (cvivit_trainer.py:107-111) must split cleanly'
    assert out['ema_deepcopy_ok'], 'EMA(vae): copy.deepcopy(PackedModule) must give independent parameters under the same keys'
    assert out['ema_update_drops_packed_copies'], 'EMA.update writes .data (no version bump): dropin must invalidate the packed copies'
    assert 'no CPU fallback' in out['cvivit_first_step'], out['cvivit_first_step']      # the first train_step reaches the HIP path and fails LOUDLY on a CPU host
    assert out['checkpoint_keys_ok'] and out['wrong_class_refused'] and out['phenaki_opt_covers_maskgit']
    checks = {(c[0], c[1]): (c[2], c[3]) for c in out['beartype_checks'] if c[3].startswith('phenaki_pytorch_amd')}
    assert checks[('CViViTTrainer.__init__', 'vae')] == ('phenaki_pytorch_amd.cvivit.CViViT',) * 2
    assert checks[('PhenakiTrainer.__init__', 'phenaki')] == ('phenaki_pytorch_amd.phenaki.Phenaki',) * 2


def test_dropin_refuses_to_run_after_the_reference_was_imported(monkeypatch):
    import types
    from phenaki_pytorch_amd import dropin
    monkeypatch.setitem(sys.modules, 'phenaki_pytorch', types.ModuleType('phenaki_pytorch'))
    with pytest.raises(RuntimeError, match='before the first'):
        dropin.install()
