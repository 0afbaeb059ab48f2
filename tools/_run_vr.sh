#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 4 5; do echo "tests PK_VOCAB_RESIDENT=$v"; PK_VOCAB_RESIDENT=$v timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x -k "vocab" 2>&1 | tail -3; done
for v in 1 2 4 5; do echo "PK_VOCAB_RESIDENT=$v"; PK_VOCAB_RESIDENT=$v timeout 300 python tools/vocab_bench.py 2>&1 | grep bf16 | grep -v "bf16x3\|Philox"; done
