"""s_memtime timeline of the n = 576 attention kernel (instrumented build):
    PK_ALT_SRC=attn bash tools/build_alt.sh atl -DPK_TIMELINE && PK_LIB_PATH=tools/_bin/libphenaki_atl.so python tools/attn_timeline.py
Per key tile of wave 0 of a few workgroups: cycles in  wait(vmcnt) | barrier | DMA issue | QK^T MFMAs | softmax | PV MFMAs."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402
from phenaki_pytorch_amd.attention import ContinuousPositionBias  # noqa: E402

torch.manual_seed(0)
S, h, n = int(os.environ.get('S', 16)), 8, 576
cpb = ContinuousPositionBias(dim=64, heads=h, num_dims=3).cuda()
tab = cpb.table(9, 8, 8)
Qp = (torch.randn(S * h * n * 64) * 0.35).cuda().to(torch.bfloat16)
Kp = (torch.randn(S * h * n * 64) * 0.35).cuda().to(torch.bfloat16)
Vt = torch.randn(S * h * n * 64).cuda().to(torch.bfloat16)
o = torch.empty(S * n, h * 64, device='cuda', dtype=torch.bfloat16)
bound = 0.35 * 0.35 * 64 * 3 + tab[4]
use_tab = os.environ.get('TAB', '1') != '0'
call = lambda: L.attn_fwd(L.BF16, Qp, Kp, Vt, o, S, h, n, n, 0, bias_table=tab if use_tab else None, score_bound=bound)
for _ in range(3):
    call()
torch.cuda.synchronize()
dbg = ctypes.CDLL(L.LIB_PATH)
N = 8 * 6 * 16
buf = (ctypes.c_ulonglong * N)()
dbg.pk_debug_attn_timeline(buf, N, 1)
call()
assert dbg.pk_debug_attn_timeline(buf, N, 0) == 0
for wg in range(8):
    rows = [[buf[(wg * 16 + t) * 6 + s] for s in range(6)] for t in range(9)]
    k = [buf[(wg * 16 + 15) * 6 + s] for s in range(2)]
    if not rows[0][0]:
        continue
    print(f'workgroup slot {wg}: entry -> first wait {rows[0][0] - k[0]}, loop {rows[-1][5] - rows[0][0]}, kernel-to-loop-end {k[1] - k[0]}')
    for t, r in enumerate(rows):
        print(f'   tile {t}: wait {r[1] - r[0]:5d} | barrier {r[2] - r[1]:5d} | issue {r[3] - r[2]:5d} | qk {r[4] - r[3]:5d} | softmax+pv {r[5] - r[4]:5d} | total {(rows[t + 1][0] if t + 1 < 9 else r[5]) - r[0]:5d}')
