#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 1 2; do echo "tests PK_VOCAB_RESIDENT=$v"; PK_VOCAB_RESIDENT=$v timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x -k "vocab" 2>&1 | tail -4; done
for v in 1 2 0; do echo "PK_VOCAB_RESIDENT=$v"; PK_VOCAB_RESIDENT=$v timeout 300 python tools/vocab_bench.py 2>&1 | grep bf16 | grep -v "bf16x3\|Philox"; done
for v in 2304 1152; do echo "M=$v"; for r in 1 2 0; do PK_VOCAB_RESIDENT=$r timeout 300 python tools/vocab_bench.py $v 2>&1 | grep "bf16 " | grep "FAST hash noise  " | sed "s/^/  resident=$r /"; done; done
