"""cProfile of the host side of the training step (bench.py's train_step leg): where the Python time of the ~1300 launches per step goes.
python tools/host_profile_train.py [dtype]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models, synthetic_context  # noqa: E402
import phenaki_pytorch_amd as P  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
cv, mg, cr, ph = build_models(mode, True)
B = 8
ctx = synthetic_context(B, 12, 768, seed=1).cuda()
ids = torch.randint(0, 65536, (B, 9, 8, 8)).cuda()
params = [p for p in list(mg.parameters()) + list(cr.parameters())]
opt = P.get_optimizer(params, lr=1e-4, wd=1e-2)


def step():
    opt.zero_grad(set_to_none=True)
    loss = ph(video_codebook_ids=ids, text_embeds=ctx)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t_issue = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 5
print(f'host issue time per step {t_issue * 1e3:.1f} ms, wall per step {t_all * 1e3:.1f} ms')
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
