"""Micro-benchmark of the epilogue flavours of pk_gemm_ex on the FF1 / to_out shapes (run on the MI355X): plain, GEGLU, LayerNorm fold with
in-loop statistics, fold with handed-over statistics, residual + bf16 copy (+ stats_out).  Each case: 20 launches captured in a hipGraph,
replayed, timed with events; median of 7 rounds, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402

REP = 20


def graph_of(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
    return g


def main():
    L.load()
    bf = torch.bfloat16
    for M in (4608, 9216):
        K, N, D = 512, 2736, 512
        x = torch.randn(M, K, device='cuda').to(bf)
        W1 = (torch.randn(N, K, device='cuda') / K ** 0.5).to(bf)
        s, t = torch.randn(N, device='cuda'), torch.randn(N, device='cuda')
        h = torch.empty(M, N // 2, device='cuda', dtype=bf)
        hp = torch.empty(M, N, device='cuda', dtype=bf)
        stats = torch.rand(M, K // 32, 2, device='cuda') + 16
        o = torch.randn(M, 512, device='cuda').to(bf)
        Wo = (torch.randn(D, 512, device='cuda') / 512 ** 0.5).to(bf)
        res = torch.randn(M, D, device='cuda')
        out = torch.empty(M, D, device='cuda')
        out_t = torch.empty(M, D, device='cuda', dtype=bf)
        st_out = torch.empty(M, D // 32, 2, device='cuda')
        hm = torch.randn(M, 1368, device='cuda').to(bf)
        W2 = torch.zeros(D, 1408, device='cuda', dtype=bf)
        W2[:, :1368] = (torch.randn(D, 1368, device='cuda') / 37).to(bf)
        cases = {
            'FF1 plain (no act)': lambda: L.gemm(L.BF16, x, W1, M, N, K, C=hp),
            'FF1 GEGLU': lambda: L.gemm(L.BF16, x, W1, M, N, K, C=h, act=L.ACT_GEGLU),
            'FF1 GEGLU + LN in-loop stats': lambda: L.gemm(L.BF16, x, W1, M, N, K, C=h, act=L.ACT_GEGLU, ln=(s, t, 1e-5)),
            'FF1 GEGLU + LN handed stats': lambda: L.gemm(L.BF16, x, W1, M, N, K, C=h, act=L.ACT_GEGLU, ln=(s, t, 1e-5), ln_stats=stats),
            'FF1 no act + LN handed stats': lambda: L.gemm(L.BF16, x, W1, M, N, K, C=hp, ln=(s, t, 1e-5), ln_stats=stats),
            'to_out + res': lambda: L.gemm(L.BF16, o, Wo, M, D, 512, C=out, res=res),
            'to_out + res + C2': lambda: L.gemm(L.BF16, o, Wo, M, D, 512, C=out, res=res, C2=out_t),
            'to_out + res + C2 + stats_out': lambda: L.gemm(L.BF16, o, Wo, M, D, 512, C=out, res=res, C2=out_t, stats_out=st_out),
            'to_out plain': lambda: L.gemm(L.BF16, o, Wo, M, D, 512, C=out),
            'FF2 + res + C2': lambda: L.gemm(L.BF16, hm, W2, M, D, 1368, C=out, res=res, C2=out_t),
        }
        graphs = {k: graph_of(f) for k, f in cases.items()}
        times = {k: [] for k in cases}
        for _ in range(7):
            for k, g in graphs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                e1.synchronize()
                times[k].append(e0.elapsed_time(e1) * 1e3 / REP)
        for k, v in times.items():
            print(f'M={M:5d}  {k:34s} {sorted(v)[len(v) // 2]:7.2f} us')


if __name__ == '__main__':
    main()
