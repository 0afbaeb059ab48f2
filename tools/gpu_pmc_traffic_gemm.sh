#!/bin/bash
# fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of one GEMM shape for a list of variants:  bash tools/gpu_pmc_traffic_gemm.sh M N K "50 24"
set -u
export TMPDIR=/tmp
for v in $4; do for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_traffic_gemm_$1_$2_$3_v${v}_$c
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o p -- python tools/one_gemm.py $1 $2 $3 $v 12 > /dev/null 2>&1
  python tools/pmc_sq.py $out
done; done
