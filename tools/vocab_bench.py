"""the fused vocabulary head (pk_vocab_sample) at the sampling shape, by mode: where its time goes -- main loop vs the gumbel-noise epilogue
vs the softmax statistics.   python tools/vocab_bench.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402
from phenaki_pytorch_amd.attention import pack_linear_weight  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
V, D = 65536, 512
torch.manual_seed(0)
A32 = torch.randn(M, D, device='cuda')
W = torch.randn(V, D, device='cuda') * 0.05
b = torch.zeros(V, device='cuda')
flops = 2.0 * M * V * D


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


for name, dt in (('bf16', L.BF16), ('bf16x3', L.BF16X3)):
    A = A32.to(L.tdtype(dt)) if L.tdtype(dt) != torch.float32 else A32
    Wp = pack_linear_weight(W, dt)
    part = torch.empty((5 * L.vocab_ntiles(V) * M,), device='cuda', dtype=torch.float32)
    U = None
    spec = L.TorchPhilox(torch.device('cuda'), M * V)
    rows = [
        ('argmax only (no noise, no lse)', lambda: L.vocab_sample(dt, A, Wp, b, M, V, D, 1.0, None, None, 1, False, part, no_noise=True)),
        ('argmax + lse (no noise)', lambda: L.vocab_sample(dt, A, Wp, b, M, V, D, 1.0, None, None, 1, True, part, no_noise=True)),
        ('FAST hash noise', lambda: L.vocab_sample(dt, A, Wp, b, M, V, D, 1.0, None, None, 1, False, part)),
        ('FAST hash noise + lse', lambda: L.vocab_sample(dt, A, Wp, b, M, V, D, 1.0, None, None, 1, True, part)),
        ('torch-Philox noise + lse', lambda: L.vocab_sample_philox(dt, A, Wp, b, M, V, D, 1.0, None, spec, True, part)),
    ]
    for label, fn in rows:
        us = timeit(fn)
        print(f'{name:7s} M={M} {label:34s} {us:8.1f} us  {flops / us / 1e6:7.1f} TF', flush=True)
