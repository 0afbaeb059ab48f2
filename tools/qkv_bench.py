"""pk_qkv_project at the sampling loop's shapes (n = 576, 8 heads, K = 512), graph-replayed; PK_QKV_TM=1|2 picks 64- / 128-row tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402

torch.manual_seed(0)
h, n, D = 8, 576, 512
wq = (torch.randn(h * 64, D, device='cuda') / D ** 0.5).to(torch.bfloat16)
wkv = (torch.randn(2 * h * 64, D, device='cuda') / D ** 0.5).to(torch.bfloat16)
qs, ks = torch.ones(64, device='cuda'), torch.ones(64, device='cuda')
sq = torch.randn(h * 64, device='cuda')
for S in (16, 8, 4):
    x = torch.randn(S * n, D, device='cuda').to(torch.bfloat16)
    nq_pad, nk_pad = L.attn_pads(n, n, 0)
    Qp = torch.empty(S * h * nq_pad * 64, device='cuda', dtype=torch.bfloat16)
    Kp = torch.empty(S * h * nk_pad * 64, device='cuda', dtype=torch.bfloat16)
    Vt = torch.empty(S * h * nk_pad * 64, device='cuda', dtype=torch.bfloat16)
    fn = lambda: L.qkv_project(x, x, wq, wkv, S, n, h, D, qs, ks, 8.0, Qp, Kp, Vt, nq_pad, nk_pad, q_ln_s=sq)
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    ts.sort()
    chk = float(Qp.float().sum() + Kp.float().sum()) , float(Vt.float().abs().sum())
    print(f'PK_QKV_TM={os.environ.get("PK_QKV_TM", "auto")}  S={S:2d}: {ts[3]:6.2f} us ({2.0 * S * n * D * 1536 / ts[3] / 1e6:.0f} TFLOP/s)  checksums {chk[0]:.3f} {chk[1]:.1f}')
