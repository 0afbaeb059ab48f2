#!/bin/bash
# the module-level GPU parity tests under every non-default setting of the tuning knobs (each knob must stay a valid configuration)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for kv in PK_LN_FOLD_FF=0 PK_LN_FOLD_FF=1 PK_LN_FOLD=0 PK_ATTN_FIXED=0 PK_CFG_SHARED_PREFIX=0 PK_BIAS_TABLE=0 PK_QKV_ATTN=0 PK_CROSS_FUSED=0 PK_ATTN_LDS_QF=2 PK_LN_FOLD_FF_MAX_ROWS=0; do
  env $kv timeout 600 python -m pytest tests/test_modules_gpu.py -m gpu -q -p no:cacheprovider -x > gpurun_out/knob_$kv.log 2>&1
  echo "$kv rc=$? $(tail -1 gpurun_out/knob_$kv.log)"
done
