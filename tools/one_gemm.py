"""One GEMM shape, N launches (for rocprofv3 PMC passes):  python tools/one_gemm.py M N K [variant] [launches] [geglu|res]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402

M, N, K = (int(a) for a in sys.argv[1:4])
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 10
mode = sys.argv[6] if len(sys.argv) > 6 else ''
L.load()
Kp = (K + 63) // 64 * 64
A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
W = torch.zeros(N, Kp, device='cuda', dtype=torch.bfloat16)
W[:, :K] = (torch.randn(N, K, device='cuda') / K ** 0.5).to(torch.bfloat16)
kw = {}
if mode == 'geglu':
    C = torch.empty(M, N // 2, device='cuda', dtype=torch.bfloat16)
    kw = dict(act=L.ACT_GEGLU)
elif mode == 'res':
    C = torch.empty(M, N, device='cuda')
    kw = dict(res=torch.randn(M, N, device='cuda'))
else:
    C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
for _ in range(launches):
    L.gemm(L.BF16, A, W, M, N, K, C=C, variant=variant, **kw)
torch.cuda.synchronize()
