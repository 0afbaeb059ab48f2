"""run a few launches of selected pk_gemm variants on one shape (target for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L
M, N, K = [int(x) for x in sys.argv[1:4]]
variants = [int(v) for v in sys.argv[4:]]
L.load()
Kp = (K + 63) // 64 * 64
A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
W = torch.zeros(N, Kp, device='cuda', dtype=torch.bfloat16)
W[:, :K] = (torch.randn(N, K, device='cuda') / K ** 0.5).to(torch.bfloat16)
RES = os.environ.get('GEMM_ONE_RES', '0') == '1'      # f32 output + f32 residual (the to_out / FF2 epilogue)
C = torch.empty(M, N, device='cuda', dtype=torch.float32 if RES else torch.bfloat16)
res = torch.randn(M, N, device='cuda') if RES else None
for v in variants:
    for _ in range(5):
        L.gemm(L.BF16, A, W, M, N, K, C=C, res=res, variant=v)
torch.cuda.synchronize()
