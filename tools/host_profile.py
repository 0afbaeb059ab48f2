"""Where does the host time of an eager (non-graph) sampling call go?  cProfile of Phenaki.sample at B = 1 (launch-bound).
    python tools/host_profile.py [B]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cv, mg, cr, ph = bench.build_models('bf16', True)
ctx = bench.synthetic_context(B, 12, 768, seed=1).cuda()
ph.encode_texts = lambda texts, output_device=None: ctx[:len(texts)]
call = lambda: ph.sample(texts=['x'] * B, num_frames=17, cond_scale=5.)
call(); call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    call()
torch.cuda.synchronize()
print(f'B={B}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per eager sample call')
pr = cProfile.Profile()
pr.enable()
call()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
