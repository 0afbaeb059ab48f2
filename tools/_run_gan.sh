#!/bin/bash
# GAN branch: parity tests, the cvivit_gan_step bench leg, and its rocprofv3 kernel stats
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 600 python -m pytest tests/test_gan_gpu.py -m gpu -q -p no:cacheprovider --tb=short > gpurun_out/gan.log 2>&1; tail -6 gpurun_out/gan.log
timeout 600 python bench.py --no-cpu --no-parity-mode --no-kernels --no-sample --legs cvivit_gan_step > gpurun_out/gan_bench_line.json 2> gpurun_out/gan_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_full.json'))
print(json.dumps(d.get('cvivit_gan_step'), indent=1))
PY
P=gpurun_out/prof_gan; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o gan -- python bench.py --no-cpu --no-parity-mode --no-kernels --no-sample --legs cvivit_gan_step > /dev/null 2> $P/err.log; echo "prof rc=$?"
find $P -name "*kernel_trace.csv" -delete; find $P -name "*.db" -delete
f=$(find $P -name "*kernel_stats.csv" | head -1); head -30 "$f" | cut -c1-200
