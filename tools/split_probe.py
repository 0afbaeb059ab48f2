"""Tile-quantisation experiments on the encode / sampling GEMMs, host side only (pk_gemm_ex's `variant` override + row-range splits):

    python tools/split_probe.py > gpurun_out/split_probe.txt

At M = 4 608 the 128 x 128 FF1 grid is 792 tiles on 512 slots (1.55 "rounds"), at M = 9 216 1 584 (3.09): does finishing the ragged last round
with smaller tiles (a second launch over the last rows) beat one launch?  Same operands, same epilogues, us per call (hipGraph of 24, median of 7).
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models  # noqa: E402
from phenaki_pytorch_amd import _lib as L  # noqa: E402
from phenaki_pytorch_amd.attention import linear_weight  # noqa: E402

torch.set_grad_enabled(False)
dt, D, REPS = L.BF16, 512, 24
cv = build_models('bf16', False)[0]
peg, att, _, ff = cv.enc_spatial_transformer.layers[1]
wo = linear_weight(att.to_out, dt)
w1g, s1, t1, _ = ff._packed_folded(dt)
w1p, w2p, ip = ff._packed(dt)
eps = ff[0].eps


def timed(fn):
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
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / REPS)
    return statistics.median(ts[2:])


for M in (() if os.environ.get('PK_PROBE_SKIP_SPLIT', '0') == '1' else (4608, 9216)):
    x = torch.randn(M, D, device='cuda')
    xt = x.to(torch.bfloat16)
    o = torch.randn(M, D, device='cuda').to(torch.bfloat16)
    y, yt = torch.empty(M, D, device='cuda'), torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
    st = torch.zeros(M, 16, 2, device='cuda')
    hm = torch.empty(M, ip, device='cuda', dtype=torch.bfloat16)
    z, zt = torch.empty(M, D, device='cuda'), torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
    L.gemm(dt, o, wo, M, D, D, C=y, res=x, C2=yt, stats_out=st)           # real statistics for the folded FF1
    fold = M <= 6144                                                       # (attention.py: the fold is used up to PK_LN_FOLD_FF_MAX_ROWS rows)
    xn = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
    L.layernorm(y, ff[0].weight, ff[0].bias, M, D, out=xn, eps=eps)

    def ff1(r0, r1, variant):
        m = r1 - r0
        if fold:
            L.gemm(dt, yt[r0:r1], w1g, m, 2 * ip, D, C=hm[r0:r1], act=L.ACT_GEGLU, ln=(s1, t1, eps), ln_stats=st[r0:r1], variant=variant)
        else:
            L.gemm(dt, xn[r0:r1], w1p, m, 2 * ip, D, C=hm[r0:r1], act=L.ACT_GEGLU, variant=variant)

    def ff2(r0, r1, variant):
        L.gemm(dt, hm[r0:r1], w2p, r1 - r0, D, ip, C=z[r0:r1], res=y[r0:r1], C2=zt[r0:r1], variant=variant)

    def tout(r0, r1, variant):
        L.gemm(dt, o[r0:r1], wo, r1 - r0, D, D, C=y[r0:r1], res=x[r0:r1], C2=yt[r0:r1], stats_out=st[r0:r1], variant=variant)

    print(f'# M = {M} rows; FF1 = {M} x {2 * ip} x 512 {"LayerNorm-folded" if fold else "plain"} + GEGLU, FF2 = {M} x 512 x {ip} + residual + bf16 copy, to_out = {M} x 512 x 512')
    ref = hm.clone()
    for name, fn, variants in (('ff1', ff1, (24, 8) if fold else (24, 9, 27, 8)), ('ff2', ff2, (8, 27, 33, 24)), ('to_out', tout, (8, 27, 24))):
        base = None
        for v in variants:
            try:
                us = timed(lambda: fn(0, M, v))
            except RuntimeError as e:
                print(f'{name:7s} variant {v:3d}: refused ({e})')
                continue
            base = base or us
            print(f'{name:7s} variant {v:3d} one launch            {us:7.2f} us')
        big, small = variants[0], 8
        if name != 'ff1':
            continue
        # row-range split: [0, r) with the 128 x 128 tiles, [r, M) with 64 x 64 tiles in a second launch
        NT = (2 * ip + 127) // 128
        for slots_rounds in (1, 2, 3):
            rt = slots_rounds * 512 // NT                              # row tiles of 128 that fill `slots_rounds` rounds of 512 slots
            if rt * 128 >= M:
                continue
            for smallv in ((8,) if fold else (8, 27)):
                r = rt * 128
                us = timed(lambda: (fn(0, r, big), fn(r, M, smallv)))
                print(f'{name:7s} split at row {r:5d} ({rt * NT} tiles of 128^2 = {rt * NT / 512:.2f} rounds) + variant {smallv} on the last {M - r} rows   {us:7.2f} us')
        for r in (M // 2,):
            us = timed(lambda: (fn(0, r, big), fn(r, M, big)))
            print(f'{name:7s} two launches of 128^2 split at row {r} (control: the cost of a second launch)   {us:7.2f} us')
    torch.cuda.synchronize()


