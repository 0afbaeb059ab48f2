#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "scatter or handoff or layernorm_fold" > gpurun_out/r2_tests9a.log 2>&1; echo "new tests rc=$?"
tail -5 gpurun_out/r2_tests9a.log
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2_tests9.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|Error|rel " gpurun_out/r2_tests9.log | tail -12
for i in 1 2; do
for v in 0 2; do
PK_LN_FOLD_FF=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-kernels --no-graph --groups 5 > gpurun_out/r2_bench9_$v.json 2> gpurun_out/r2_bench9_$v.err
python - <<PY
import json
d=json.load(open('gpurun_out/r2_bench9_$v.json')); print('LN_FOLD_FF=$v', 'encode ms', d['ms_per_step'], 'decode', d['decode'].get('ms_per_step'), 'sample', d['sample']['seconds_by_launch_mode'], 'cfg3', d['sample_cfg3']['seconds_per_sample_call'], 'mv', d['make_video']['wall_clock_s'])
PY
done
done
