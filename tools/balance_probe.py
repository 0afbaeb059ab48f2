"""Upper bound on what in-kernel stream-K could buy on the hot path's short-K GEMMs, measured instead of estimated: the SAME kernel on the same
operand shapes at row counts whose tile count fills the resident workgroup slots exactly (balanced) and at the hot path's row counts (ragged).  If T(ragged) is on
the line through the balanced points, tile quantisation costs nothing that a redistribution of k-iterations could recover; the distance above the line is the most
stream-K could save BEFORE its own fix-up traffic (a 16-64 KB f32 slab per split tile + a last-arriver epilogue, 5-13 us per seam in the guide's price list).

    python tools/balance_probe.py [--iters 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402

# (name, N, K, epilogue, tile rows, tile cols, resident slots, [row counts: balanced ... ragged hot-path ones])
CASES = [
    ('to_out 64x64 tiles (5 WG/CU)', 512, 512, 'res', 64, 64, 1280, [2048, 4096, 4608, 5120, 8192, 9216, 10240]),
    ('FF2 64x64 tiles', 512, 1368, 'res', 64, 64, 1280, [2048, 4096, 4608, 5120, 8192]),
    ('FF1 GEGLU 128x128 tiles (2 WG/CU)', 2736, 512, 'geglu', 128, 128, 512, [1472, 2944, 4608, 5888, 9216, 11776]),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--rounds', type=int, default=5)
    args = ap.parse_args()
    L.load()
    for name, N, K, epi, tr, tc, slots, rows in CASES:
        Kp = (K + 63) // 64 * 64
        W = torch.zeros(N, Kp, device='cuda', dtype=torch.bfloat16)
        W[:, :K] = (torch.randn(N, K, device='cuda') / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device='cuda')
        setups = []
        for M in rows:
            A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
            if epi == 'geglu':
                C = torch.empty(M, N // 2, device='cuda', dtype=torch.bfloat16)
                kw = dict(bias=bias, act=L.ACT_GEGLU)
            else:
                C = torch.empty(M, N, device='cuda')
                kw = dict(res=torch.randn(M, N, device='cuda'), C2=torch.empty(M, N, device='cuda', dtype=torch.bfloat16))
            L.gemm(L.BF16, A, W, M, N, K, C=C, **kw)
            setups.append((M, A, C, kw))
        torch.cuda.synchronize()
        times = {M: [] for M, *_ in setups}
        for _ in range(args.rounds):
            for M, A, C, kw in setups:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    L.gemm(L.BF16, A, W, M, N, K, C=C, **kw)
                e1.record()
                torch.cuda.synchronize()
                times[M].append(e0.elapsed_time(e1) / args.iters * 1e3)
        print(f'== {name}: N = {N}, K = {K}, variant {L.load().pk_gemm_auto_variant(1, 0, rows[2], N, K, K, Kp, rows[2])} at M = {rows[2]}')
        for M, *_ in setups:
            t = sorted(times[M])[len(times[M]) // 2]
            tiles = -(-M // tr) * -(-N // tc)
            print(f'   M = {M:6d}  tiles {tiles:5d} = {tiles / slots:5.2f} x {slots} slots, {tiles / 256:5.2f} per CU   {t:7.2f} us   {t / tiles * 1e3:7.2f} ns per tile   '
                  f'{2.0 * M * N * K / t / 1e6:6.0f} TF', flush=True)


if __name__ == '__main__':
    main()
