"""Quick parity check of experimental pk_gemm main-loop variants against variant 24 (the 128 x 128 LDS-DMA loop) and an f64 reference:
    python tools/gemm_check.py 50,70 [--reps 3]
Shapes cover M / N / K tails, 1 / 2 / odd numbers of k-tiles, and a multi-round grid (race screen: repeated, compared bit for bit)."""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402

L.load()
variants = [int(v) for v in sys.argv[1].split(',')]
reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 3
X3 = '--x3' in sys.argv                                  # split-bf16 operands (f32 A rows, host-split W planes) instead of bf16
DT = L.BF16X3 if X3 else L.BF16
SHAPES = [(300, 200, 96), (1000, 520, 1368), (4608, 512, 512), (129, 2736, 512), (2100, 1160, 64), (256, 256, 128), (513, 257, 192), (4096, 4096, 4096), (9216, 2736, 512)]
bad = 0
for M, N, K in SHAPES:
    q = 32 if X3 else 64
    Kp = (K + q - 1) // q * q
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn(M, K, device='cuda', generator=g)
    W32 = torch.zeros(N, Kp, device='cuda')
    W32[:, :K] = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    if X3:
        W = L.split_planes(W32)
        Wref = W32
    else:
        A, W = A.to(torch.bfloat16), W32.to(torch.bfloat16)
        Wref = W
    bias = torch.randn(N, device='cuda', generator=g)
    res = torch.randn(M, N, device='cuda', generator=g)
    ref = (A.double() @ Wref[:, :K].double().t() + bias.double() + res.double()).float() if M * N * K < 2e10 else None
    C0 = torch.empty(M, N, device='cuda')
    L.gemm(DT, A, W, M, N, K, C=C0, bias=bias, res=res, variant=24)
    for v in variants:
        for r in range(reps):
            C = torch.full((M, N), float('nan'), device='cuda')
            L.gemm(DT, A, W, M, N, K, C=C, bias=bias, res=res, variant=v)
            same = torch.equal(C, C0)
            err = (C - ref).abs().max().item() / ref.abs().max().item() if ref is not None else float('nan')
            ok = same or (ref is not None and err < (4e-5 if X3 else 3e-5))
            bad += not ok
            if r == 0 or not ok:
                print(f'{M}x{N}x{K} v{v} rep{r}: bit-identical to v24 {same}  rel err vs f64 {err:.2e}  {"ok" if ok else "MISMATCH"}', flush=True)
print('FAILED' if bad else 'all ok')
sys.exit(1 if bad else 0)
