#!/bin/bash
# round 5, call F: knob re-check for the sampler (feed-forward LayerNorm fold beyond 6144 rows), same box, alternating
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for r in 1 2; do for v in 6144 100000; do
PK_LN_FOLD_FF_MAX_ROWS=$v timeout 300 python bench.py --no-cpu --no-parity-mode --no-kernels --groups 5 --legs sample,sample_b32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FOLD_FF_MAX_ROWS=$v encode', round(d['value']), 'sample', d['sample']['value'], d['sample']['ms'], 'legs', d['legs'])"
done; done
date
