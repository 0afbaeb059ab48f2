"""Wall time of the Phenaki training step (bench.py's train_step body: forward + backward + AdamW, BASELINE geometry, B = 8) for a same-box A/B of two
trees:   python tools/train_time.py [--root DIR] [--mode bf16x3] [--steps 20] [--rounds 3]
--root: a directory holding another build of the package + bench.py (e.g. an export of the previous commit); default: this tree."""
import argparse
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument('--root', default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap.add_argument('--mode', default='bf16x3')
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--loss', action='store_true', help='print the loss after every round (numerics of the two trees side by side)')
args = ap.parse_args()
args.root = os.path.abspath(args.root)
sys.path.insert(0, args.root)
os.chdir(args.root)

import torch  # noqa: E402
from bench import build_models, synthetic_context  # noqa: E402
import phenaki_pytorch_amd as P  # noqa: E402

assert os.path.abspath(P.__file__).startswith(os.path.abspath(args.root)), P.__file__
B = 8
cv, mg, cr, ph = build_models(args.mode, True)
for m in (mg, cr):
    m.train()
ctx = synthetic_context(B, 12, 768, seed=1).cuda()
g = torch.Generator(device='cpu')
g.manual_seed(4)
ids = torch.randint(0, 65536, (B, 9, 8, 8), generator=g).cuda()
params = [p for p in list(mg.parameters()) + list(cr.parameters()) if p.requires_grad]
opt = P.get_optimizer(params, lr=1e-4, wd=1e-2)
torch.manual_seed(0)


def step():
    with torch.enable_grad():
        opt.zero_grad(set_to_none=True)
        loss = ph(video_codebook_ids=ids, text_embeds=ctx)
        loss.backward()
    opt.step()
    return loss


for _ in range(4):
    step()
torch.cuda.synchronize()
times = []
for r in range(args.rounds):
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) / args.steps * 1e3)
    if args.loss:
        print(f'  round {r}: loss {float(loss):.6f}')
print(f'{args.root} {args.mode}: ms per step ' + ' '.join(f'{t:.2f}' for t in times) + f'  (min {min(times):.2f})')
