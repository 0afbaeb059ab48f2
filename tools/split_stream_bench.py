"""Does running the two halves of an encode batch on two HIP streams (inside one captured graph) fill the gaps a single dependent chain of
one-wave-of-tiles kernels leaves?  B = 8 as 1 x 8 / 2 x 4 / 4 x 2, graph replay timing.   python tools/split_stream_bench.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cv, _, _, _ = bench.build_models('bf16', False)
video = bench.synthetic_video(8, 17, 256, 0).cuda()


def encode_split(parts):
    if parts == 1:
        return [cv(video, return_only_codebook_ids=True)]
    cur = torch.cuda.current_stream()
    outs, streams = [], [torch.cuda.Stream() for _ in range(parts)]
    chunk = 8 // parts
    for i, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(cv(video[i * chunk:(i + 1) * chunk], return_only_codebook_ids=True))
    for s in streams:
        cur.wait_stream(s)
    return outs


ref = None
for parts in (1, 2, 4, 1, 2):
    replay, out = bench.capture(lambda: encode_split(parts))
    ids = torch.cat(out, dim=0)
    if ref is None:
        ref = ids.clone()
    same = torch.equal(ids, ref)
    for _ in range(5):
        replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    print(f'{parts} stream(s) x batch {8 // parts}: {statistics.median(ts):.4f} ms per 8 videos  (ids equal: {same})', flush=True)
