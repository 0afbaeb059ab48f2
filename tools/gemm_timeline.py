"""s_memtime timeline of the LDS-DMA GEMM main loop (instrumented build):
    bash tools/build_alt.sh tl -DPK_TIMELINE && PK_LIB_PATH=tools/_bin/libphenaki_tl.so python tools/gemm_timeline.py M N K variant
Per k-tile of wave 0 of a few workgroups: cycles spent in  wait(vmcnt) | barrier | DMA issue | ds_read + MFMA | (next iteration)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402

M, N, K, variant = (int(a) for a in sys.argv[1:5])
lib = L.load()
Kp = (K + 63) // 64 * 64
A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
W = torch.zeros(N, Kp, device='cuda', dtype=torch.bfloat16)
W[:, :K] = (torch.randn(N, K, device='cuda') / K ** 0.5).to(torch.bfloat16)
mode = sys.argv[5] if len(sys.argv) > 5 else ''
kw = {}
C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
if mode == 'res':
    C = torch.empty(M, N, device='cuda')
    kw = dict(res=torch.randn(M, N, device='cuda'))
for _ in range(3):
    L.gemm(L.BF16, A, W, M, N, K, C=C, variant=variant, **kw)
torch.cuda.synchronize()
dbg = ctypes.CDLL(L.LIB_PATH)
dbg.pk_debug_timeline_clear()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
L.gemm(L.BF16, A, W, M, N, K, C=C, variant=variant, **kw)
e1.record()
torch.cuda.synchronize()
print(f'launch {e0.elapsed_time(e1) * 1e3:.1f} us (events, includes launch overhead)')
buf = (ctypes.c_ulonglong * (8 * 5 * 40))()
assert dbg.pk_debug_timeline(buf, 8 * 5 * 40) == 0
nt = min((K + 63) // 64, 38)
t0 = min(v for v in buf if v)
for wg in range(8):
    rows = [[buf[(wg * 40 + kt) * 5 + s] for s in range(5)] for kt in range(nt)]
    if not rows[0][0]:
        continue
    print(f'workgroup slot {wg}: start +{rows[0][0] - t0} ticks')
    for kt, r in enumerate(rows):
        nxt = rows[kt + 1][0] if kt + 1 < nt else r[4]
        print(f'   kt {kt:2d}: wait {r[1] - r[0]:6d} | barrier {r[2] - r[1]:6d} | issue {r[3] - r[2]:6d} | mfma {r[4] - r[3]:6d} | total {nxt - r[0]:6d}')
    k = [buf[(wg * 40 + 39) * 5 + s] for s in range(3)]
    print(f'   entry -> first wait {rows[0][0] - k[0]} | main loop {rows[-1][4] - rows[0][0]} | epilogue (stores issued + vmcnt 0) {k[2] - k[1]} | kernel {k[2] - k[0]} ticks')
