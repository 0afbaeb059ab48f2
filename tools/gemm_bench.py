"""A/B micro-benchmark of the pk_gemm main-loop variants on the hot path's real shapes, beside a YARDSTICK that never touches the
product: torch.mm on the same operands (hipBLASLt / rocBLAS through ATen).  Run on the MI355X:

    python tools/gemm_bench.py [--iters 20] [--mode bf16|bf16x3|f32] [--variants 8,24,...] [--out gpurun_out/gemm_bench.json]

Interleaved rounds in ONE process (cdna guide rule 24), random operands (rule 25); prints TFLOP/s (median over rounds) per
shape x variant.  `torch.mm` is tools-only: nothing in phenaki_pytorch_amd/ calls a BLAS library."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402

SHAPES = [  # (M, N, K, note)
    (4608, 512, 512, 'tokenizer q / out (B=8)'),
    (4608, 1024, 512, 'tokenizer kv'),
    (4608, 2736, 512, 'tokenizer FF1 (GEGLU)'),
    (4608, 512, 1368, 'tokenizer FF2'),
    (4096, 512, 6144, 'patch embed'),
    (9216, 512, 512, 'maskgit q / out (2B=16 x 576)'),
    (9216, 2736, 512, 'maskgit FF1'),
    (9216, 512, 1368, 'maskgit FF2'),
    (18432, 512, 512, 'tokenizer q / out (B=32)'),
    (18432, 2736, 512, 'tokenizer FF1 (B=32)'),
    (36864, 2736, 512, 'maskgit FF1 (B=32)'),
    (4608, 65536, 512, 'vocab head as plain GEMM'),
    (8192, 8192, 8192, 'square 8k (kernel ceiling)'),
]
VARIANTS = {'bf16': {8: 'd64s2', 9: 'd128s2', 24: 'd128w8s2', 27: 'd128x64w4', 33: 'pc64 4+2 s3'},
            'bf16x3': {8: 'd64s2', 24: 'd128w8s2', 27: 'd128x64w4', 9: 'd128s2', 3: 'd64s4'},
            'f32': {8: 'd64s2', 24: 'd128w8s2', 9: 'd128s2', 3: 'd64s4'}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--mode', default='bf16', choices=['bf16', 'bf16x3', 'f32'])
    ap.add_argument('--variants', type=str, default='', help='comma list, default all of the mode')
    ap.add_argument('--shapes', type=str, default='', help='comma list of shape indices, default all')
    ap.add_argument('--res', action='store_true', help='add bias + f32 residual (the out-projection epilogue)')
    ap.add_argument('--geglu', action='store_true', help='GEGLU epilogue with a T (bf16) output of N / 2 columns: what the FF1 calls of the hot path run (even N only)')
    ap.add_argument('--splitk256', type=int, default=0, help='also time pk_gemm_splitk(tile=2) with this many K-slices + pk_sum_batch (bf16, f32 C)')
    ap.add_argument('--no-yardstick', action='store_true')
    ap.add_argument('--out', default='gpurun_out/gemm_bench.json')
    args = ap.parse_args()
    L.load()
    variants = dict(VARIANTS[args.mode])
    if args.variants:
        keep = [int(v) for v in args.variants.split(',')]
        variants = {v: variants.get(v, f'v{v}') for v in keep}
    shapes = SHAPES if not args.shapes else [SHAPES[int(i)] for i in args.shapes.split(',')]
    dt = {'bf16': L.BF16, 'bf16x3': L.BF16X3, 'f32': L.F32}[args.mode]
    q = 64 if args.mode == 'bf16' else 32
    out = {}
    for M, N, K, note in shapes:
        Kp = (K + q - 1) // q * q
        A32 = torch.randn(M, K, device='cuda')
        W32 = torch.zeros(N, Kp, device='cuda')
        W32[:, :K] = torch.randn(N, K, device='cuda') / K ** 0.5
        if args.mode == 'bf16':
            A, W = A32.to(torch.bfloat16), W32.to(torch.bfloat16)
        elif args.mode == 'bf16x3':
            A, W = A32, L.split_planes(W32)
        else:
            A, W = A32, W32
        big = (M * N) >= (1 << 28)
        C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16 if (big and args.mode == 'bf16') else torch.float32)
        kw = {}
        if args.geglu and N % 2 == 0 and args.mode in ('bf16', 'bf16x3'):
            C = torch.empty(M, N // 2, device='cuda', dtype=torch.bfloat16 if args.mode == 'bf16' else torch.float32)
            kw = dict(bias=torch.randn(N, device='cuda'), act=L.ACT_GEGLU)
        if args.res and C.dtype == torch.float32:
            kw = dict(bias=torch.randn(N, device='cuda'), res=torch.randn(M, N, device='cuda'))
        live = {}
        for v in variants:
            try:
                L.gemm(dt, A, W, M, N, K, C=C, variant=v, **kw)      # warm; variants that do not exist for this mode / build are dropped
                live[v] = variants[v]
            except RuntimeError:
                pass
        # yardstick: torch.mm with the operand types the product kernel sees (bf16 x bf16 -> bf16 / f32 x f32 -> f32), W pre-transposed
        # both ways; the better one is reported
        yard = {}
        if not args.no_yardstick:
            Ay = A32.to(torch.bfloat16) if args.mode == 'bf16' else A32
            Wy = W32[:, :K].contiguous().to(Ay.dtype)
            Wt = Wy.t().contiguous()
            yard = {'mm(A, W^T view)': lambda: torch.mm(Ay, Wy.t()), 'mm(A, Wt contiguous)': lambda: torch.mm(Ay, Wt)}
            for f in yard.values():
                f()
        # split-K on the 256 x 256 two-group loop (pk_gemm_splitk tile = 2 + pk_sum_batch): for shapes with few tiles and a long K
        if args.mode == 'bf16' and args.splitk256 and C.dtype == torch.float32 and not kw and K % (64 * args.splitk256) == 0:
            part = torch.empty(args.splitk256, M * N, device='cuda')

            def sk(part=part):
                L.gemm_splitk(dt, A, W, M, N, K, args.splitk256, part, tile=2)
                L.sum_batch(part, args.splitk256, C, M * N)
            sk()
            if K <= 8192 and M * N <= (1 << 22):
                ref = (A.double() @ W[:, :K].double().t()).float()
                assert (C - ref).abs().max() <= 3e-5 * ref.abs().max(), 'split-K 256 result differs from the f64 product'
            yard = dict(yard)
            yard[f'splitK{args.splitk256} x 256^2 + sum'] = sk
        torch.cuda.synchronize()
        times = {k: [] for k in list(live) + list(yard)}
        iters = max(2, args.iters // 8) if M * N * K > 1e11 else args.iters
        for _ in range(args.rounds):
            for k in times:
                fn = (lambda k=k: L.gemm(dt, A, W, M, N, K, C=C, variant=k, **kw)) if k in live else yard[k]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / iters * 1e-3)
        row = {}
        for k, ts in times.items():
            ts.sort()
            t = ts[len(ts) // 2]
            row[live.get(k, k)] = dict(us=t * 1e6, tflops=2.0 * M * N * K / t / 1e12)
        out[f'{M}x{N}x{K} {note}'] = row
        print(f'{M}x{N}x{K:5d} {note:32s} ' + '  '.join(f'{k}: {r["us"]:7.1f}us {r["tflops"]:6.0f}TF' for k, r in row.items()), flush=True)
        del A, W, A32, W32, C
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(dict(mode=args.mode, res=args.res, shapes=out), open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
