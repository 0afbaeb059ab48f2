#!/bin/bash
# round 5, call E: the LDS-slab PEG + one-launch patch-embed finish: parity, micro A/B, encode / sample A/B; the two fixed dist tests
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests"; date
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dist_gpu.py -m gpu -q -p no:cacheprovider -k "peg or patch_embed or rccl_executes or data_parallel_training" > gpurun_out/r5e_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5e_tests.log
PK_PEG_SEQ=2 timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "peg" > gpurun_out/r5e_peg2.log 2>&1; echo "peg (forced slab kernel) rc=$?"; tail -3 gpurun_out/r5e_peg2.log
echo "== PEG micro A/B"; date
for v in 0 1 0 1; do PK_PEG_SEQ=$v timeout 120 python tools/peg_ab.py 2>/dev/null; done
echo "== encode A/B (PK_PEG_SEQ x PK_PATCH_FINISH_ONE)"; date
for r in 1 2; do for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg
  PK_PEG_SEQ=$1 PK_PATCH_FINISH_ONE=$2 timeout 200 python bench.py --encode-only --groups 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PEG_SEQ=$1 FINISH_ONE=$2', round(d['value']), 'frames/s', round(d['ms_per_step'],4), 'ms')"
done; done
echo "== sample A/B"; date
for v in 0 1 0 1; do PK_PEG_SEQ=$v timeout 300 python bench.py --no-cpu --no-parity-mode --no-kernels --groups 5 --legs sample,sample_cfg3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PEG_SEQ=$v sample', d['sample']['value'], d['sample']['ms'], 'cfg3', d['legs'].get('sample_cfg3'))"; done
date
