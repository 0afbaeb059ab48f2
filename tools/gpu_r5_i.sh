#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/parity.jsonl
timeout 600 python -m pytest tests/test_gan_gpu.py -m gpu -q -p no:cacheprovider -k "generator_gan_step_matches_reference" > gpurun_out/r5i_gan.log 2>&1; echo "gan rc=$?"; tail -8 gpurun_out/r5i_gan.log
grep adaptive gpurun_out/parity.jsonl | cut -c1-700
cp gpurun_out/parity.jsonl gpurun_out/parity_gan_r5i.jsonl
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_line.json 2> gpurun_out/bench_full.err; echo "bench rc=$?"
cp gpurun_out/bench_full.json gpurun_out/bench_driver_full.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_driver_full.json'))
print('encode', round(d['value']), round(d['ms_per_step'],4), 'hbm', d['roofline_hbm'])
print('x3 roofline', d['parity_mode']['roofline']['kernel'], d['parity_mode']['roofline']['traffic'])
PY
