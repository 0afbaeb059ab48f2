"""A/B micro-benchmark of the pk_gemm main-loop variants on the hot path's real shapes (run on the MI355X):

    python tools/gemm_bench.py [--iters 30]

Interleaved rounds in ONE process (cdna guide rule 24); prints TFLOP/s (median over rounds) per shape x variant."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402

SHAPES = [  # (M, N, K, note)
    (4608, 512, 512, 'tokenizer q / out'),
    (4608, 1024, 512, 'tokenizer kv'),
    (4608, 2736, 512, 'tokenizer FF1 (GEGLU)'),
    (4608, 512, 1368, 'tokenizer FF2'),
    (4096, 512, 6144, 'patch embed'),
    (9216, 512, 512, 'maskgit q / out (2B=16 x 576)'),
    (9216, 2736, 512, 'maskgit FF1'),
    (9216, 512, 1368, 'maskgit FF2'),
    (4608, 65536, 512, 'vocab head as plain GEMM'),
]
VARIANTS = {8: 'd64s2', 9: 'd128s2', 24: 'd128w8s2', 27: 'd128x64w4', 33: 'pc64 4+2 s3'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--variants', type=str, default='', help='comma list, default all')
    ap.add_argument('--res', action='store_true', help='add bias + f32 residual (the out-projection epilogue)')
    args = ap.parse_args()
    L.load()
    global VARIANTS
    if args.variants:
        VARIANTS = {8: 'd64s2', 9: 'd128s2', 24: 'd128w8s2', 33: 'pc64 4+2 s3'}
    out = {}
    for M, N, K, note in SHAPES:
        Kp = (K + 63) // 64 * 64
        A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
        W = torch.zeros(N, Kp, device='cuda', dtype=torch.bfloat16)
        W[:, :K] = (torch.randn(N, K, device='cuda') / K ** 0.5).to(torch.bfloat16)
        C = torch.empty(M, N, device='cuda', dtype=torch.float32 if N < 60000 else torch.bfloat16)
        kw = {}
        if args.res and C.dtype == torch.float32:
            kw = dict(bias=torch.randn(N, device='cuda'), res=torch.randn(M, N, device='cuda'))
        times = {v: [] for v in VARIANTS}
        for v in VARIANTS:
            L.gemm(L.BF16, A, W, M, N, K, C=C, variant=v, **kw)      # warm
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for v in VARIANTS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    L.gemm(L.BF16, A, W, M, N, K, C=C, variant=v, **kw)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / args.iters * 1e-3)
        row = {}
        for v, ts in times.items():
            ts.sort()
            t = ts[len(ts) // 2]
            row[VARIANTS[v]] = dict(us=t * 1e6, tflops=2.0 * M * N * K / t / 1e12)
        out[f'{M}x{N}x{K} {note}'] = row
        print(f'{M}x{N}x{K:5d} {note:32s} ' + '  '.join(f'{k}: {r["us"]:7.1f}us {r["tflops"]:6.0f}TF' for k, r in row.items()), flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/gemm_bench.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
