#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do
for v in 1 0; do
PK_CROSS_FUSED=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-kernels --no-graph > gpurun_out/r2_bench6_$v.json 2> gpurun_out/r2_bench6_$v.err
python - <<PY
import json
d=json.load(open('gpurun_out/r2_bench6_$v.json')); print('CROSS_FUSED=$v', 'encode ms', round(d['ms_per_step'],4), 'sample', d['sample']['seconds_by_launch_mode'], 'cfg3', d['sample_cfg3']['seconds_per_sample_call'], 'mv', d['make_video']['wall_clock_s'])
PY
done
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode > gpurun_out/r2_bench6k.json 2> gpurun_out/r2_bench6k.err
