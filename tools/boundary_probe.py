"""Decomposition of the in-graph vs standalone gap of the encode step's kernels:

    python tools/boundary_probe.py [--rows 4608] > gpurun_out/boundary_probe.txt

One spatial layer of the BASELINE C-ViViT at B = 8 (M = 4 608 rows): qkv_attn -> to_out -> FF1(GEGLU) -> FF2, the library's own launches with
the model's packed weights.  Every scenario is captured as ONE hipGraph of REPS iterations and timed between two HIP events (median of 7
replays); what is printed is microseconds per iteration.
  hot        one kernel re-launched on the SAME buffers (the DESIGN 4.1 "standalone" column: inputs in this XCD's L2 from the last launch)
  rot3/rot8  the same kernel rotating over 3 / 8 buffer sets (180 MB: beyond the 32 MB of L2, inside the Infinity Cache; 480 MB: beyond it)
  chain      the four kernels in their real dependency order on one buffer set (what the encode graph runs) and its pairs
  dirty X    to_out behind a fill kernel that leaves X MB dirty in an UNRELATED buffer (boundary + write-back of X)
  coldA/R    to_out behind a copy kernel that re-writes its A operand / its residual (operand arrives from another CU's write-back)
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models  # noqa: E402
from phenaki_pytorch_amd import _lib as L  # noqa: E402
from phenaki_pytorch_amd.attention import _full_bias, linear_weight  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=4608)
ap.add_argument('--reps', type=int, default=24)
args = ap.parse_args()
torch.set_grad_enabled(False)
M, D, REPS = args.rows, 512, args.reps
S, n, h = M // 64, 64, 8
dt = L.BF16

cv = build_models('bf16', False)[0]
peg, att, _, ff = cv.enc_spatial_transformer.layers[1]
wq, sq, tq = att._folded_q(dt)
wkv, wo = linear_weight(att.to_kv, dt), linear_weight(att.to_out, dt)
w1g, s1, t1, _ = ff._packed_folded(dt)
w1p, w2p, ip = ff._packed(dt)
bias = _full_bias(cv.spatial_rel_pos_bias(8, 8))
eps = ff[0].eps
NSET = 8
dev = 'cuda'


def mk(shape, dtype):
    return [torch.randn(shape, device=dev).to(dtype) for _ in range(NSET)]


x, xt = mk((M, D), torch.float32), mk((M, D), torch.bfloat16)
o = mk((M, D), torch.bfloat16)
y, yt = mk((M, D), torch.float32), mk((M, D), torch.bfloat16)
st = [torch.zeros((M, 16, 2), device=dev) for _ in range(NSET)]
hm = mk((M, ip), torch.bfloat16)
z, zt = mk((M, D), torch.float32), mk((M, D), torch.bfloat16)
spare = [torch.empty((64 << 20,), device=dev, dtype=torch.uint8) for _ in range(2)]        # unrelated buffers for the dirty / copy kernels
o_src, x_src = torch.randn((M, D), device=dev).to(torch.bfloat16), torch.randn((M, D), device=dev)


def k_qkv(r):
    L.qkv_attn(xt[r], xt[r], wq, wkv, S, n, h, D, att.q_scale, att.k_scale, float(att.scale), o[r], bias=bias, q_ln_s=sq)


def k_out(r, c2=True):
    L.gemm(dt, o[r], wo, M, D, D, C=y[r], res=x[r], C2=yt[r] if c2 else None, stats_out=st[r] if c2 else None)


def k_ff1(r):
    L.gemm(dt, yt[r], w1g, M, 2 * ip, D, C=hm[r], act=L.ACT_GEGLU, ln=(s1, t1, eps), ln_stats=st[r])


def k_ff2(r):
    L.gemm(dt, hm[r], w2p, M, D, ip, C=z[r], res=y[r], C2=zt[r])


def fill(mb):
    nbytes = int(mb * (1 << 20))
    return lambda r: spare[0][:nbytes].fill_(1) if nbytes else None


def timed(seq, nsets=1, label=''):
    """seq: list of callables f(r); one iteration = the whole list on buffer set (it % nsets)"""
    def body():
        for it in range(REPS):
            for f in seq:
                f(it % nsets)
    for f in seq:
        f(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        body()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / REPS)
    us = statistics.median(ts[2:])
    if label:
        print(f'{label:58s} {us:8.2f} us / iteration', flush=True)
    return us


for r in range(NSET):                      # every buffer set holds a real layer's intermediates (row statistics included)
    k_qkv(r); k_out(r); k_ff1(r); k_ff2(r)
torch.cuda.synchronize()
print(f'# boundary probe: M = {M} rows, one C-ViViT spatial layer, bf16; us per iteration (hipGraph of {REPS} iterations, median of 7 replays)')
kern = dict(qkv_attn=k_qkv, to_out=k_out, ff1=k_ff1, ff2=k_ff2)
hot, rot3, rot8 = {}, {}, {}
for name, f in kern.items():
    hot[name] = timed([f], 1, f'hot   {name}')
for name, f in kern.items():
    rot3[name] = timed([f], 3, f'rot3  {name} (inputs beyond L2, inside the Infinity Cache)')
for name, f in kern.items():
    rot8[name] = timed([f], NSET, f'rot8  {name} (inputs beyond the Infinity Cache)')
to_out_plain = timed([lambda r: k_out(r, c2=False)], 1, 'hot   to_out without the bf16 copy + row statistics')

print()
chain = timed([k_qkv, k_out, k_ff1, k_ff2], 1, 'chain qkv_attn -> to_out -> ff1 -> ff2 (one buffer set)')
chain3 = timed([k_qkv, k_out, k_ff1, k_ff2], 3, 'chain, rotating over 3 buffer sets')
print(f'      sum of the four hot launches {sum(hot.values()):8.2f} us -> the chain costs {chain - sum(hot.values()):+.2f} us more ({(chain / sum(hot.values()) - 1) * 100:+.1f} %)')
for a, b in (('qkv_attn', 'to_out'), ('to_out', 'ff1'), ('ff1', 'ff2')):
    p = timed([kern[a], kern[b]], 1)
    print(f'pair  {a:9s} -> {b:9s} {p:8.2f} us = hot sum {hot[a] + hot[b]:6.2f} {p - hot[a] - hot[b]:+6.2f}')
# the consumer of a pair behind an INDEPENDENT producer (same kernels, the consumer reads another buffer set: same boundary, operands not fresh)
for a, b in (('qkv_attn', 'to_out'), ('to_out', 'ff1'), ('ff1', 'ff2')):
    p = timed([lambda r, a=a: kern[a](0), lambda r, b=b: kern[b](1)], 1)
    print(f'indep {a:9s} || {b:9s} {p:8.2f} us (consumer reads set 1, producer writes set 0)  {p - hot[a] - hot[b]:+6.2f} vs hot sum')

print()
print('# to_out behind a fill kernel that leaves X MB dirty in an unrelated buffer: T([fill, to_out]) - T([fill])')
for mb in (0.004, 4.7, 9.4, 14.1, 28.2, 56.4):
    f = fill(mb)
    alone = timed([f], 1)
    pair = timed([f, k_out], 1)
    print(f'dirty {mb:6.1f} MB: fill alone {alone:6.2f}  fill + to_out {pair:6.2f}  -> to_out behind it {pair - alone:6.2f} us (hot {hot["to_out"]:.2f})')

print()
print('# to_out behind a copy kernel that re-writes one of its operands (4.7 MB bf16 A / 9.4 MB f32 residual) vs the same copy into a spare buffer')
sp_bf = spare[1][:M * D * 2].view(torch.bfloat16).view(M, D)
sp_f32 = spare[1][:M * D * 4].view(torch.float32).view(M, D)
for label, dst_real, dst_spare, src in (('A (attention output, bf16)', o[0], sp_bf, o_src), ('residual (f32)', x[0], sp_f32, x_src)):
    c_real = lambda r, d=dst_real, s_=src: d.copy_(s_)
    c_spare = lambda r, d=dst_spare, s_=src: d.copy_(s_)
    a0 = timed([c_spare], 1)
    p_spare = timed([c_spare, k_out], 1)
    p_real = timed([c_real, k_out], 1)
    print(f'cold  {label:28s}: copy alone {a0:6.2f}  copy(spare) + to_out {p_spare:6.2f}  copy(operand) + to_out {p_real:6.2f}  -> fresh operand costs {p_real - p_spare:+6.2f} us')
