#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/census
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/census -o c -- python tools/train_census.py bf16x3 6 2>&1 | tail -2
python tools/train_census.py --summary gpurun_out/census 6 | tee gpurun_out/train_census.txt
find gpurun_out/census -name "*kernel_trace.csv" -delete; find gpurun_out/census -name "*.db" -delete
