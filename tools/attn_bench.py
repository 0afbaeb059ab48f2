"""time pk_attn_prep + pk_attn_fwd on the hot path's shapes (run on the MI355X):  python tools/attn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L  # noqa: E402

SHAPES = [  # S, h, nq, n_kv, nnull, bias, causal, note
    (16, 8, 576, 576, 0, True, False, 'maskgit self-attn (2B=16)'),
    (16, 8, 576, 12, 2, False, False, 'maskgit cross-attn (L=12)'),
    (72, 8, 64, 64, 0, True, False, 'tokenizer spatial (B*T=72)'),
    (512, 8, 9, 9, 0, False, True, 'tokenizer temporal (B*hw=512)'),
]


def timeit(fn, iters=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    L.load()
    dt, td = L.BF16, torch.bfloat16
    for S, h, nq, n_kv, nnull, has_bias, causal, note in SHAPES:
        q = torch.randn(S * nq, h * 64, device='cuda')
        kv = torch.randn(S * n_kv, 2 * h * 64, device='cuda')
        null_kv = torch.randn(h, 2 * max(nnull, 1), 64, device='cuda')
        qs, ks = torch.ones(64, device='cuda'), torch.ones(64, device='cuda')
        nq_pad, nk_pad = L.attn_pads(nq, n_kv, nnull)
        Qp = torch.empty(S * h * nq_pad * 64, device='cuda', dtype=td)
        Kp = torch.empty(S * h * nk_pad * 64, device='cuda', dtype=td)
        Vt = torch.empty(S * h * nk_pad * 64, device='cuda', dtype=td)
        O = torch.empty(S * nq, h * 64, device='cuda', dtype=td)
        bias = torch.randn(h, nq, n_kv, device='cuda') if has_bias else None
        slopes = torch.rand(h, device='cuda') if causal else None
        t_prep = timeit(lambda: L.attn_prep(dt, q, kv, null_kv, qs, ks, 8.0, Qp, Kp, Vt, S, h, nq, n_kv, nnull))
        t_fwd = timeit(lambda: L.attn_fwd(dt, Qp, Kp, Vt, O, S, h, nq, n_kv, nnull, bias=bias, slopes=slopes, causal=causal))
        flops = 4.0 * S * h * nq * (n_kv + nnull) * 64
        print(f'{note:34s} prep {t_prep:7.1f} us   fwd {t_fwd:7.1f} us   {flops / t_fwd / 1e6:7.1f} TFLOP/s   (PK_ATTN_MAX_QF={os.environ.get("PK_ATTN_MAX_QF", "4")})', flush=True)


if __name__ == '__main__' and len(sys.argv) == 1:
    main()


def small():
    """the fused short-sequence kernel on the tokenizer shapes"""
    L.load()
    for S, n, causal, has_bias, note in [(72, 64, False, True, 'spatial'), (512, 9, True, False, 'temporal')]:
        h = 8
        q = torch.randn(S * n, h * 64, device='cuda')
        kv = torch.randn(S * n, 2 * h * 64, device='cuda')
        qs, ks = torch.ones(64, device='cuda'), torch.ones(64, device='cuda')
        bias = torch.randn(h, n, n, device='cuda') if has_bias else None
        slopes = torch.rand(h, device='cuda') if causal else None
        O = torch.empty(S * n, h * 64, device='cuda', dtype=torch.bfloat16)
        t = timeit(lambda: L.attn_small(q, kv, qs, ks, 8.0, O, S, h, n, bias=bias, slopes=slopes, causal=causal))
        print(f'attn_small {note:10s} S={S} n={n}: {t:7.1f} us', flush=True)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'small':
    small()
