"""Which source lines of the training step issue ATen kernels (fills, copies, casts, cats) -- the launches the step's own C ABI does not account
for.  torch.profiler with stacks over 2 steps; every ATen op that launched a device kernel is attributed to the innermost frame inside the package.
    python tools/train_aten_census.py [dtype]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models, synthetic_context  # noqa: E402
import phenaki_pytorch_amd as P  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
cv, mg, cr, ph = build_models(mode, True)
for m in (mg, cr):
    m.train()
B = 8
ctx = synthetic_context(B, 12, 768, seed=1).cuda()
g = torch.Generator(device='cpu')
g.manual_seed(4)
ids = torch.randint(0, 65536, (B, 9, 8, 8), generator=g).cuda()
params = [p for p in list(mg.parameters()) + list(cr.parameters()) if p.requires_grad]
opt = P.get_optimizer(params, lr=1e-4, wd=1e-2)


def step():
    with torch.enable_grad():
        opt.zero_grad(set_to_none=True)
        loss = ph(video_codebook_ids=ids, text_embeds=ctx)
        loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()

by_site = collections.Counter()
by_op = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::'):
        continue
    if not ev.kernels:                                       # only the ops that launched something themselves
        continue
    site = 'outside the package (autograd engine / optimizer)'
    for fr in ev.stack:
        if 'phenaki_pytorch_amd' in fr or 'bench.py' in fr:
            site = fr.split('phenaki_pytorch_amd/')[-1]
            break
    by_site[(ev.name, site)] += len(ev.kernels)
    by_op[ev.name] += len(ev.kernels)
print(f'# ATen-launched device kernels per step ({mode}, {STEPS} steps profiled)')
for k, v in by_op.most_common():
    print(f'{v / STEPS:7.1f}  {k}')
print('# by call site')
for (name, site), v in by_site.most_common(60):
    print(f'{v / STEPS:7.1f}  {name:28s} {site}')
