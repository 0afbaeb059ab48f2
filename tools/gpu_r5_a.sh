#!/bin/bash
# round 5, call A: the new benched-config parity tests, the boundary probe, a same-box baseline of the bench legs
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
echo "== benched-config parity tests"; date
timeout 900 python -m pytest tests/test_benched_configs_gpu.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r5_benched_tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r5_benched_tests.log
cp gpurun_out/parity.jsonl gpurun_out/parity_benched.jsonl 2>/dev/null
echo "== boundary probe"; date
timeout 300 python tools/boundary_probe.py > gpurun_out/boundary_probe.txt 2> gpurun_out/boundary_probe.err; echo "probe rc=$?"
cat gpurun_out/boundary_probe.txt; tail -5 gpurun_out/boundary_probe.err
echo "== baseline bench (encode + sample legs)"; date
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-kernels --groups 9 --legs decode,sample,sample_cfg3 > gpurun_out/r5_base_line.json 2> gpurun_out/r5_base.err; echo "bench rc=$?"
tail -1 gpurun_out/r5_base_line.json | cut -c1-1500
date
