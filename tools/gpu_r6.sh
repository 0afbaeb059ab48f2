#!/bin/bash
# round-6 GPU legs, one sub-command per gpurun call:  bash tools/gpu_r6.sh <leg> [args]
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
leg=${1:-p8}; shift || true
case $leg in
p8)   # 8-phase 256x256 main loop: parity cases, then the yardstick table
  timeout 600 python tools/gemm_check.py ${3:-50} 2>&1 | tail -30
  timeout 900 python tools/gemm_bench.py --mode bf16 --variants ${1:-24,50} --shapes ${2:-4,6,9,10,11,12} --out gpurun_out/gemm_p8.json ${4:-} 2>&1 | tee gpurun_out/gemm_p8.txt
  ;;
x3)   # the same loop on the split-bf16 (hi | lo) operand images: parity vs variant 24, then the table
  timeout 600 python tools/gemm_check.py ${3:-50} --x3 2>&1 | tail -30
  timeout 900 python tools/gemm_bench.py --mode bf16x3 --variants ${1:-9,24,50} --shapes ${2:-2,6,9,10,11,12} --out gpurun_out/gemm_p8_x3.json ${4:-} 2>&1 | tee gpurun_out/gemm_p8_x3.txt
  ;;
vendor)  # which Tensile kernels torch.mm picks (yardstick only)
  cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/vendor -o vendor -- python $GRAFT_REPO_ROOT/tools/vendor_mm.py 2>&1 | grep -v "^W2\|rocprofv3" | tee $GRAFT_REPO_ROOT/gpurun_out/vendor_mm.txt
  cd $GRAFT_REPO_ROOT && python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/vendor/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open('gpurun_out/vendor_kernels.txt', 'w') as o:
        for r in rows:
            o.write(f"{r['Calls']:>6} calls  avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name']}\n")
    print(open('gpurun_out/vendor_kernels.txt').read()[:6000])
PY
  ;;
esac
