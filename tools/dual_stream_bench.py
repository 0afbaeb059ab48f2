"""Does splitting the tokenizer batch over TWO HIP streams (two independent half-batches captured as parallel branches of
one hipGraph) hide the inter-kernel bubbles and tails of the single-stream pass?  (run on the MI355X)

    python tools/dual_stream_bench.py [--batch 8] [--steps 50]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--splits', type=int, default=2)
    args = ap.parse_args()
    from oracle import weights
    from oracle.configs import FULL
    from tests.util import load_product
    cv = load_product('full', FULL, device='cuda', dtype='bf16', with_critic=False)[0]
    B, NS = args.batch, args.splits
    video = weights.synthetic_video(B, 17, 256, 256, seed=0).cuda()
    parts = [p.contiguous() for p in video.chunk(NS, dim=0)]
    ref = cv.tokenize(video)
    for p in parts:
        cv.tokenize(p)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(NS)]

    def run_split():
        cur = torch.cuda.current_stream()
        outs = []
        for st, p in zip(streams, parts):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(cv.tokenize(p))
        for st in streams:
            cur.wait_stream(st)
        return outs

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    # warm both shapes on the side streams (allocator pools, packed weights), then capture
    run_split()
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        ids1 = cv.tokenize(video)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        ids2 = run_split()
    g1.replay(); g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(ids1, ref)
    assert torch.equal(torch.cat(ids2, 0), ref), 'split batches must give the same token ids'
    for rnd in range(3):
        t1, t2 = timed(g1.replay), timed(g2.replay)
        print(f'round {rnd}: one stream B={B}: {t1:.4f} ms/step   {NS} streams x B={B // NS}: {t2:.4f} ms/step', flush=True)


if __name__ == '__main__':
    main()
