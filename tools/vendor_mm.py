"""Yardstick only (nothing in the product links a BLAS): run torch.mm (hipBLASLt / rocBLAS through ATen) on the three shapes VERDICT r5 names, so that
`rocprofv3 --kernel-trace --stats -- python tools/vendor_mm.py` records which Tensile kernels the vendor library picks for them -- the kernel NAMES spell
out macro-tile (MT), wave tiling (WG / MIWT), MFMA shape (MI), DirectToLds (DTL), depthU (DU), prefetch (PGR / PLR), LDS buffers (1LDSB), workgroup
mapping (WGM / WGMXCC) and stream-K (SK).  profiles/gemm_vendor_r06.txt is the digest."""
import torch

SHAPES = [(8192, 8192, 8192), (4608, 65536, 512), (4096, 512, 6144), (9216, 2736, 512), (36864, 2736, 512)]
for M, N, K in SHAPES:
    A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    W = (torch.randn(N, K, device='cuda') / K ** 0.5).to(torch.bfloat16)
    Wt = W.t().contiguous()
    for _ in range(3):
        torch.mm(A, W.t())
        torch.mm(A, Wt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, fn in (('A @ W^T (view)', lambda: torch.mm(A, W.t())), ('A @ Wt', lambda: torch.mm(A, Wt))):
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(f'{M}x{N}x{K} {name}: {t * 1e6:.1f} us {2.0 * M * N * K / t / 1e12:.0f} TF', flush=True)
    del A, W, Wt
