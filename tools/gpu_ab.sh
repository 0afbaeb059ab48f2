#!/bin/bash
# same-box A/B of two builds of the C ABI: default in-tree .so vs $1
set -u
ALT=$1
export TMPDIR=/tmp
for i in 1 2; do
for v in base alt; do
if [ $v = alt ]; then export PK_LIB_PATH=$ALT; else unset PK_LIB_PATH; fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-kernels --no-graph --groups 7 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
python - <<PY
import json
d=json.load(open('gpurun_out/ab_$v.json')); print('$v', 'encode ms', round(d['ms_per_step'],4), 'decode', round(d['decode']['ms_per_step'],4), 'sample', round(d['sample']['seconds_per_sample_call']*1e3,2), 'cfg3', round(d['sample_cfg3']['seconds_per_sample_call']*1e3,2))
PY
done
done
unset PK_LIB_PATH
echo "--- gemm_bench base"; python tools/gemm_bench.py --iters 20 --rounds 3 2>&1 | grep -v amdgpu | cut -c1-150
echo "--- gemm_bench alt"; PK_LIB_PATH=$ALT python tools/gemm_bench.py --iters 20 --rounds 3 2>&1 | grep -v amdgpu | cut -c1-150
