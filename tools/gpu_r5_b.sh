#!/bin/bash
# round 5, call B: LFQ auxiliary-loss kernels + GAN objective with it, advisor-fix tests, the ping-pong GEMM (parity first, then speed), tile-split probe
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
echo "== LFQ aux + GAN + tokenizer training tests"; date
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "lfq" > gpurun_out/r5b_lfq.log 2>&1; echo "lfq rc=$?"; tail -4 gpurun_out/r5b_lfq.log
timeout 1200 python -m pytest tests/test_gan_gpu.py tests/test_train_cvivit_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r5b_gan.log 2>&1; echo "gan rc=$?"; tail -12 gpurun_out/r5b_gan.log
cp gpurun_out/parity.jsonl gpurun_out/parity_gan_r5b.jsonl 2>/dev/null
echo "== ping-pong GEMM parity"; date
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "pingpong or main_loop_variants" > gpurun_out/r5b_pp.log 2>&1; echo "pp rc=$?"; tail -6 gpurun_out/r5b_pp.log
echo "== split / ping-pong probe"; date
timeout 300 python tools/split_probe.py > gpurun_out/split_probe.txt 2> gpurun_out/split_probe.err; echo "probe rc=$?"
cat gpurun_out/split_probe.txt; tail -3 gpurun_out/split_probe.err
date
