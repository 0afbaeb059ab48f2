"""N Phenaki training steps (bench.py's train_step body, BASELINE geometry, B = 8) for a rocprofv3 --kernel-trace --stats census:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/census -o c -- python tools/train_census.py bf16x3 6
    python tools/train_census.py --summary gpurun_out/census 6"""
import csv
import glob
import os
import sys

if sys.argv[1] == '--summary':
    # steady state only: every step is preceded by a marker launch (torch.cuda._sleep); the first two marked steps (optimizer-state
    # allocation, first-use set-up) and everything before the first marker (model construction) are left out
    import collections
    d, steps = sys.argv[2], int(sys.argv[3])
    tr = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'sleep' in r['Kernel_Name'].lower() or 'spin' in r['Kernel_Name'].lower()]
    assert len(marks) == steps, f'{len(marks)} markers for {steps} steps'
    skip = 2
    body = [r for r in rows[marks[skip]:] if not ('sleep' in r['Kernel_Name'].lower() or 'spin' in r['Kernel_Name'].lower())]
    ns = steps - skip
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in body:
        a = acc[r['Kernel_Name']]
        a[0] += 1
        a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot = sum(v[1] for v in acc.values())
    n = sum(v[0] for v in acc.values())
    print(f'# {tr}: {tot / 1e3 / ns:.2f} ms of kernel time and {n / ns:.0f} launches per step (steady state: steps {skip + 1}..{steps} of {steps})')
    for nm, (c, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:48]:
        print(f'{us / tot * 100:5.1f}%  {c / ns:7.1f} / step x {us / c:8.1f} us  {nm[:130]}')
    # the attention kernels by grid size (self-attention n = 576 vs the 14-key cross-attention share one kernel name)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in body:
        nm = r['Kernel_Name']
        if 'attn' not in nm:
            continue
        key = (nm[:60], r.get('Grid_Size_X', r.get('Grid_Size', '?')))
        acc[key][0] += 1
        acc[key][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print('# attention kernels by grid size (threads): launches per step x average us')
    for (nm, grid), (n_, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f'{n_ / ns:7.1f} / step x {us / n_:8.1f} us  grid {grid:>8}  {nm}')
    sys.exit(0)

import torch  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models, synthetic_context  # noqa: E402
import phenaki_pytorch_amd as P  # noqa: E402

mode, steps = sys.argv[1], int(sys.argv[2])
leg = sys.argv[3] if len(sys.argv) > 3 else 'phenaki'
B = 8
if leg != 'phenaki':
    # the tokenizer's steps (bench.py cvivit_train_step / cvivit_gan_step bodies): 'cvivit' reconstruction step, 'gan_gen' / 'gan_discr' the two halves of a GAN step
    from torch import nn
    from bench import BASELINE_CFG, synthetic_video
    torch.manual_seed(0)
    gan = leg != 'cvivit'
    vgg = None
    if gan:
        vgg = nn.Sequential(nn.AvgPool2d(4), nn.Flatten(), nn.Linear(3 * 64 * 64, 64), nn.Tanh(), nn.Linear(64, 32))
        for q in vgg.parameters():
            q.requires_grad_(False)
    cv = P.CViViT(use_vgg_and_gan=gan, vgg=vgg, **BASELINE_CFG['cvivit']).cuda().train()
    P.set_compute_dtype(cv, mode)
    video = synthetic_video(B, 17, 256, 5).cuda()
    if leg == 'gan_discr':
        params = [p for p in cv.discr.parameters() if p.requires_grad]
    else:
        params = [p for n, p in cv.named_parameters() if p.requires_grad and not n.startswith('discr.')]
    opt = P.get_optimizer(params, lr=1e-4, wd=0.)
    for i in range(steps):
        torch.cuda._sleep(1000)
        with torch.enable_grad():
            opt.zero_grad(set_to_none=True)
            loss = cv(video, return_discr_loss=True, apply_grad_penalty=True) if leg == 'gan_discr' else cv(video)
            loss.backward()
        opt.step()
    torch.cuda.synchronize()
    print('loss', float(loss))
    sys.exit(0)
cv, mg, cr, ph = build_models(mode, True)
for m in (mg, cr):
    m.train()
ctx = synthetic_context(B, 12, 768, seed=1).cuda()
g = torch.Generator(device='cpu')
g.manual_seed(4)
ids = torch.randint(0, 65536, (B, 9, 8, 8), generator=g).cuda()
params = [p for p in list(mg.parameters()) + list(cr.parameters()) if p.requires_grad]
opt = P.get_optimizer(params, lr=1e-4, wd=1e-2)
torch.manual_seed(0)
for i in range(steps):
    torch.cuda._sleep(1000)                                  # step marker for --summary
    with torch.enable_grad():
        opt.zero_grad(set_to_none=True)
        loss = ph(video_codebook_ids=ids, text_embeds=ctx)
        loss.backward()
    opt.step()
torch.cuda.synchronize()
print('loss', float(loss))
