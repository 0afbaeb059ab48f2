#!/bin/bash
# full GPU test suite + a short bench (no CPU / parity-mode / kernel-table legs): the routine check after a kernel change
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2_tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_tests.log | tail -12
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-kernels --no-graph --groups 5 > gpurun_out/r2_bench_quick.json 2> gpurun_out/r2_bench_quick.err
python - <<PY
import json
d=json.load(open('gpurun_out/r2_bench_quick.json')); legs=d.get('legs', {})
print('encode ms', round(d['ms_per_step'],4), 'sample', d.get('sample', {}).get('value'), 'legs', {k: v.get('value', v.get('ms_per_step')) for k, v in legs.items()})
PY
done
