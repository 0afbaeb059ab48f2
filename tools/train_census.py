"""N Phenaki training steps (bench.py's train_step body, BASELINE geometry, B = 8) for a rocprofv3 --kernel-trace --stats census:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/census -o c -- python tools/train_census.py bf16x3 6
    python tools/train_census.py --summary gpurun_out/census 6"""
import csv
import glob
import os
import sys

if sys.argv[1] == '--summary':
    d, steps = sys.argv[2], int(sys.argv[3])
    path = glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    n = sum(int(r['Calls']) for r in rows)
    print(f'# {path}: {tot / 1e6 / steps:.2f} ms of kernel time and {n / steps:.0f} launches per step ({steps} steps incl. 2 warm-up)')
    for r in rows[:45]:
        print(f"{float(r['TotalDurationNs']) / tot * 100:5.1f}%  {int(r['Calls']) / steps:7.1f} / step x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:130]}")
    # the attention kernels by grid size (self-attention n = 576 vs the 14-key cross-attention share one kernel name)
    tr = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    if tr:
        import collections
        acc = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(tr[0])):
            nm = r['Kernel_Name']
            if 'attn' not in nm:
                continue
            key = (nm[:60], r.get('Grid_Size_X', r.get('Grid_Size', '?')))
            acc[key][0] += 1
            acc[key][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print('# attention kernels by grid size (threads): launches per step x average us')
        for (nm, grid), (n_, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            print(f'{n_ / steps:7.1f} / step x {us / n_:8.1f} us  grid {grid:>8}  {nm}')
    sys.exit(0)

import torch  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models, synthetic_context  # noqa: E402
import phenaki_pytorch_amd as P  # noqa: E402

mode, steps = sys.argv[1], int(sys.argv[2])
B = 8
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
    with torch.enable_grad():
        opt.zero_grad(set_to_none=True)
        loss = ph(video_codebook_ids=ids, text_embeds=ctx)
        loss.backward()
    opt.step()
torch.cuda.synchronize()
print('loss', float(loss))
