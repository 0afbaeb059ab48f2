#!/bin/bash
# full round evidence: GPU tests (parity records), the bench line exactly as the driver runs it, smoke
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
echo "== tests"; date
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/tests_full.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/tests_full.log | tail -8
echo "== smoke"; date
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== bench"; date
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_line.json 2> gpurun_out/bench_full.err; echo "bench rc=$?"
cp gpurun_out/bench_full.json gpurun_out/bench_driver_full.json      # later profiling runs of bench.py rewrite bench_full.json
tail -c 600 gpurun_out/bench_full.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_driver_full.json'))
print('encode', round(d['value']), 'frames/s', round(d['ms_per_step'],4), 'ms; roofline', d['roofline']['kernel'], round(d['roofline']['frac'],3), 'traffic', d['roofline']['traffic'])
line=open('gpurun_out/bench_line.json').read().strip().splitlines()[-1]
c=json.loads(line)
print('compact line bytes', len(line), 'sample', c['sample']['value'], 'parity', c['parity'], 'legs', c['legs'])
print('cpu', c['cpu_baseline'])
PY
echo "== done"; date
