#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -q -p no:cacheprovider -k "qkv_attn or attention_block or cvivit or transformer or blocks" > gpurun_out/r2_tests7.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|Error|rel " gpurun_out/r2_tests7.log | tail -12
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-sample > gpurun_out/r2_bench7.json 2> gpurun_out/r2_bench7.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench7.json')); print('encode ms', round(d['ms_per_step'],4), 'decode', round(d['decode']['ms_per_step'],4))
for r in d['kernels']:
    if 'qkv_attn' in r['kernel']: print(r['leg'], r['kernel'], r['avg_us'])
PY
