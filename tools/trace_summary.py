"""summarise a rocprofv3 --kernel-trace CSV between the last two FillFunctor<double> markers of tools/trace_encode.py: launches, busy
time, idle gaps, per-kernel totals, and any kernel that is not this library's.   python tools/trace_summary.py <dir> [--list]"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'FillFunctor<double>' in r['Kernel_Name']]
seg = rows[idx[-2] + 1: idx[-1]]
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(seg, seg[1:])]
print(f'{len(seg)} launches, span {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, gaps: sum {sum(max(g, 0) for g in gaps) / 1e3:.1f} us '
      f'median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, overlapped launches {sum(g < 0 for g in gaps)}')
by = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = re.sub(r'\(.*', '', r['Kernel_Name'])[:90]
    by[n][0] += 1
    by[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    flag = '' if ('pk::' in n or n.startswith('pk_')) else '   <-- not a kernel of this library'
    print(f'{t / 1e3:10.1f} us {100 * t / busy:5.1f}%  {c:5d} x {t / c / 1e3:7.1f} us  {n}{flag}')
if '--list' in sys.argv:
    for r in seg:
        print('   ', r['Kernel_Name'][:110])
