"""Host logic of the discriminator's training graph on the CPU (phenaki_pytorch_amd/discriminator.py): the weight re-orderings, the folded
1 / sqrt(2), the transposition flags of the differentiation-closed product Function and the second-order closure behind the gradient penalty,
with the C-ABI calls it makes replaced by torch expressions (tools/emulate_discr_cpu.py) -- logits, hinge + penalty and every gradient against
oracle/gan_oracle.py.  The kernels themselves are tested on the GPU (tests/test_gan_gpu.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_discriminator_graph_composition_matches_oracle(monkeypatch):
    import emulate_discr_cpu as E
    with torch.enable_grad():
        worst = E.main(setter=monkeypatch.setattr, cases=((32, 16), ((64, 32), 4)))
    assert worst < 1e-4
