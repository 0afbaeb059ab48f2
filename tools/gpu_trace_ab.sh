#!/bin/bash
# rocprofv3 per-kernel totals of the whole bench (no cpu / parity / kernel-table legs): old tree vs working tree, same box
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/trace_ab
for which in old new; do
  if [ $which = old ]; then dir=$PWD/tools/_bin/old; else dir=$PWD; fi
  out=$PWD/gpurun_out/trace_ab/$which
  rm -rf $out; mkdir -p $out
  (cd $dir && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-kernels --no-graph --groups 3 > $out/bench.json 2> $out/err.txt)
  find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
done
python - <<'PY'
import csv, glob, collections, re
def load(which):
    f = glob.glob(f'gpurun_out/trace_ab/{which}/**/*kernel_stats.csv', recursive=True)[0]
    d = {}
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Name']).replace('void ', '').replace('pk::', '')
        name = re.sub(r',(false|true|0|1|2)>$', '>', name)
        d[name] = d.get(name, [0, 0.0]); d[name][0] += int(r['Calls']); d[name][1] += float(r['TotalDurationNs']) / 1e3
    return d
o, n = load('old'), load('new')
tot_o, tot_n = sum(v[1] for v in o.values()), sum(v[1] for v in n.values())
print(f'total kernel time: old {tot_o/1e3:.1f} ms, new {tot_n/1e3:.1f} ms')
rows = []
for k in set(o) | set(n):
    a, b = o.get(k, [0, 0.0]), n.get(k, [0, 0.0])
    rows.append((b[1] - a[1], k, a, b))
for dlt, k, a, b in sorted(rows, key=lambda r: -abs(r[0]))[:28]:
    print(f'{dlt/1e3:+8.2f} ms  {k[:70]:70s} old {a[0]:6d} x {a[1]/max(a[0],1):7.1f} us   new {b[0]:6d} x {b[1]/max(b[0],1):7.1f} us')
PY
