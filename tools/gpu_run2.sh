#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== gap"; date
timeout 600 python tools/bf16_gap.py > gpurun_out/r2_gap.log 2>&1; echo "gap rc=$?"
cat gpurun_out/r2_gap.log | grep -v amdgpu.ids
echo "== bench (kernels only)"; date
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r2_bench2.err
echo "== done"; date
