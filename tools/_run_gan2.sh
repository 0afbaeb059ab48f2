#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gan_gpu.py tests/test_train_cvivit_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x 2>&1 | tail -4
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_modules_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x -k "splitk or weight_grad or training_step or recon" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu --no-parity-mode --no-kernels --no-sample --legs cvivit_gan_step,cvivit_train_step > /dev/null 2> gpurun_out/gan_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_full.json'))
g=d.get('cvivit_gan_step'); print({k:g[k] for k in ('generator_step_ms','discriminator_step_ms','discriminator_step_no_penalty_ms','peak_memory_gb')}); print(d['cvivit_train_step']['ms_per_step'])
PY
