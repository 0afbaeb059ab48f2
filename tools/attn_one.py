"""a few launches of pk_attn_fwd on the MaskGit self-attention shape (target for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L
L.load()
S, h, nq, n_kv, nnull = 16, 8, 576, 576, 0
dt, td = L.BF16, torch.bfloat16
q = torch.randn(S * nq, h * 64, device='cuda'); kv = torch.randn(S * n_kv, 2 * h * 64, device='cuda')
qs, ks = torch.ones(64, device='cuda'), torch.ones(64, device='cuda')
nq_pad, nk_pad = L.attn_pads(nq, n_kv, nnull)
Qp = torch.empty(S * h * nq_pad * 64, device='cuda', dtype=td); Kp = torch.empty(S * h * nk_pad * 64, device='cuda', dtype=td)
Vt = torch.empty(S * h * nk_pad * 64, device='cuda', dtype=td); O = torch.empty(S * nq, h * 64, device='cuda', dtype=td)
bias = torch.randn(h, nq, n_kv, device='cuda')
L.attn_prep(dt, q, kv, None, qs, ks, 8.0, Qp, Kp, Vt, S, h, nq, n_kv, nnull)
for use_bias in (True, False):
    for _ in range(3):
        L.attn_fwd(dt, Qp, Kp, Vt, O, S, h, nq, n_kv, nnull, bias=bias if use_bias else None)
torch.cuda.synchronize()
