#!/bin/bash
# GPU box script: tests -> bench -> kernel-trace profile -> PMC traffic passes.  Logs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
echo "== tests"; date
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r2_tests1.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2_tests1.log
tail -15 gpurun_out/r2_tests1.log
echo "== bench"; date
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r2_bench1.err
head -c 1500 gpurun_out/r2_bench1.json
echo "== done"; date
