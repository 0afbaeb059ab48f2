#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
echo "== tests"; date
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2_tests3.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_tests3.log | tail -30
echo "== bench A/B fused"; date
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-sample > gpurun_out/r2_bench3a.json 2> gpurun_out/r2_bench3a.err; echo "rc=$?"
PK_QKV_ATTN=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-sample --no-kernels > gpurun_out/r2_bench3b.json 2> gpurun_out/r2_bench3b.err; echo "rc=$?"
python - <<'PY'
import json
for f in ('gpurun_out/r2_bench3a.json','gpurun_out/r2_bench3b.json'):
    try:
        d=json.load(open(f)); print(f, 'encode ms', round(d['ms_per_step'],4), 'decode ms', round(d['decode']['ms_per_step'],4))
    except Exception as e: print(f, 'ERR', e)
PY
echo "== done"; date
