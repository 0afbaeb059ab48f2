"""n = 576 self-attention (S = 16, h = 8): no bias / full (h, n, n) f32 bias / relative-position table, graph-replayed timing"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L
from phenaki_pytorch_amd.attention import ContinuousPositionBias
torch.manual_seed(0)
S, h, n = int(os.environ.get('S', 16)), 8, int(os.environ.get('N', 576))
cpb = ContinuousPositionBias(dim=64, heads=h, num_dims=3).cuda()
T = n // 64
full = cpb(T, 8, 8); tab = cpb.table(T, 8, 8)
Qp = (torch.randn(S * h * n * 64) * 0.35).cuda().to(torch.bfloat16)
Kp = (torch.randn(S * h * n * 64) * 0.35).cuda().to(torch.bfloat16)
Vt = torch.randn(S * h * n * 64).cuda().to(torch.bfloat16)
o = torch.empty(S * n, h * 64, device='cuda', dtype=torch.bfloat16)
def bench(name, fn, reps=20):
    try:
        fn(); torch.cuda.synchronize()
    except RuntimeError as e:
        print(f'{name:28s} not available ({str(e)[:40]})'); return
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort(); print(f'{name:28s} {ts[2]:8.1f} us  ({4.0 * S * h * n * n * 64 / ts[2] / 1e6:.0f} TFLOP/s)', flush=True)
bench('no bias', lambda: L.attn_fwd(L.BF16, Qp, Kp, Vt, o, S, h, n, n, 0))
bench('full f32 bias', lambda: L.attn_fwd(L.BF16, Qp, Kp, Vt, o, S, h, n, n, 0, bias=full))
bench('relative-position table', lambda: L.attn_fwd(L.BF16, Qp, Kp, Vt, o, S, h, n, n, 0, bias_table=tab))
bound = 0.35 * 0.35 * 64 * 3 + tab[4]      # comfortably above any q.k of these operands
bench('no bias, fixed offset', lambda: L.attn_fwd(L.BF16, Qp, Kp, Vt, o, S, h, n, n, 0, score_bound=bound))
bench('full f32 bias, fixed offset', lambda: L.attn_fwd(L.BF16, Qp, Kp, Vt, o, S, h, n, n, 0, bias=full, score_bound=bound))
bench('table, fixed offset', lambda: L.attn_fwd(L.BF16, Qp, Kp, Vt, o, S, h, n, n, 0, bias_table=tab, score_bound=bound))
