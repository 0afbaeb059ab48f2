#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
P=gpurun_out/prof_gan; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o gan -- python bench.py --no-cpu --no-parity-mode --no-kernels --no-sample --legs cvivit_gan_step > /dev/null 2> $P/err.log; echo "prof rc=$?"
find $P -name "*kernel_trace.csv" -delete; find $P -name "*.db" -delete
