"""Mean SQ counters per dispatch per kernel from a rocprofv3 --pmc output directory:  python tools/pmc_sq.py DIR"""
import collections
import csv
import glob
import os
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            a = agg[row["Kernel_Name"][:110]][row['Counter_Name']]
            a[0] += 1
            a[1] += float(row['Counter_Value'])
for k, cs in agg.items():
    if 'pk::' not in k:
        continue
    print(k)
    wc = cs.get('SQ_WAVE_CYCLES', [1, 0.0])
    wc = wc[1] / max(wc[0], 1)
    for c, (n, v) in sorted(cs.items()):
        m = v / n
        print(f'   {c:28s} {m:14.0f}' + (f'  {m / wc:6.1%} of WAVE_CYCLES' if wc and c.startswith('SQ_') else ''))
