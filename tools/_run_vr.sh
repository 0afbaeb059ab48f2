#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --tb=short -x -k "vocab" 2>&1 | tail -15
for v in 1 0; do echo "PK_VOCAB_RESIDENT=$v"; PK_VOCAB_RESIDENT=$v timeout 300 python tools/vocab_bench.py 2>&1 | grep bf16 | grep -v bf16x3; done
for v in 1 0 1 0; do PK_VOCAB_RESIDENT=$v timeout 300 python bench.py --no-cpu --no-parity-mode --no-kernels --legs sample --groups 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PK_VOCAB_RESIDENT=$v sample', d['sample'])"; done
