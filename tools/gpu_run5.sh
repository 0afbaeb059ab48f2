#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
echo "== tests"; date
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2_tests5.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|Error|rel " gpurun_out/r2_tests5.log | tail -20
echo "== bench"; date
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode > gpurun_out/r2_bench5a.json 2> gpurun_out/r2_bench5a.err; echo "rc=$?"
PK_QKV_ATTN=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-kernels > gpurun_out/r2_bench5b.json 2> gpurun_out/r2_bench5b.err; echo "rc=$?"
tail -3 gpurun_out/r2_bench5a.err
python - <<'PY'
import json
for f in ('gpurun_out/r2_bench5a.json','gpurun_out/r2_bench5b.json'):
    try:
        d=json.load(open(f)); print(f, 'encode ms', round(d['ms_per_step'],4), 'decode ms', round(d['decode']['ms_per_step'],4), 'sample', d['sample']['seconds_by_launch_mode'], 'cfg3', d['sample_cfg3']['seconds_per_sample_call'], 'mv', d['make_video']['wall_clock_s'])
    except Exception as e: print(f, 'ERR', e)
PY
echo "== done"; date
