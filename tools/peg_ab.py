"""time pk_peg at the hot path's shapes (hipGraph of 24 calls, median of 7): run once with PK_PEG_SEQ=0 (row kernel) and once with =1 (LDS slab kernel)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402

REPS = 24
for (B, T, H, W, D, causal) in ((8, 9, 8, 8, 512, False), (8, 9, 8, 8, 512, True), (16, 9, 8, 8, 512, False), (16, 10, 8, 8, 512, False), (64, 9, 8, 8, 512, False)):
    M = B * T * H * W
    x = torch.randn(M, D, device='cuda')
    wt, bias = torch.randn(27, D, device='cuda') * 0.1, torch.randn(D, device='cuda') * 0.1
    out, out_t = torch.empty_like(x), torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
    fn = lambda: L.peg(x, wt, bias, out, B, T, H, W, D, causal, out_t=out_t)
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / REPS)
    us = statistics.median(ts[2:])
    mb = M * D * (4 + 4 + 2) / 1e6
    print(f'PK_PEG_SEQ={os.environ.get("PK_PEG_SEQ", "1")}  ({B},{T},{H},{W},{D}) causal={int(causal)}: {us:7.2f} us  {mb / us * 1e-3:6.2f} TB/s of {mb:.1f} MB in + out')
