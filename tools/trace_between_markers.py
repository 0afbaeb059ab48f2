"""print the dispatches between the two FillFunctor<double> markers of tools/trace_encode.py in a rocprofv3 --kernel-trace output directory (csv or rocpd .db)
    python tools/trace_between_markers.py DIR"""
import csv
import glob
import os
import sqlite3
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
for f in glob.glob(os.path.join(d, '**', '*.db'), recursive=True):
    db = sqlite3.connect(f)
    rows += [(s, e, n) for n, s, e in db.execute('select name, start, end from kernels')]
    try:
        rows += [(s, e, 'MEMCPY ' + str(n)) for n, s, e in db.execute('select name, start, end from memory_copies')]
    except sqlite3.Error:
        pass
rows.sort()
marks = [i for i, r in enumerate(rows) if 'FillFunctor<double>' in r[2]]
lo, hi = marks[-2], marks[-1]
span = rows[lo + 1:hi]
print(f'{len(span)} dispatches between the markers, {sum(e - s for s, e, _ in span) / 1e3:.1f} us of kernel time, wall {(span[-1][1] - span[0][0]) / 1e3:.1f} us')
for s, e, n in span:
    print(f'{(e - s) / 1e3:8.1f} us  {n[:120]}')
