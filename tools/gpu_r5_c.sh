#!/bin/bash
# round 5, call C: ping-pong variants (parity, then speed), the GAN adaptive-weight pin
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
echo "== ping-pong GEMM parity"; date
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "pingpong or main_loop_variants or gemm_split_bf16" -x > gpurun_out/r5c_pp.log 2>&1; echo "pp rc=$?"; tail -6 gpurun_out/r5c_pp.log
echo "== ping-pong probe"; date
PK_PROBE_SKIP_SPLIT=1 timeout 400 python tools/split_probe.py > gpurun_out/pp_probe.txt 2> gpurun_out/pp_probe.err; echo "probe rc=$?"
cat gpurun_out/pp_probe.txt; tail -3 gpurun_out/pp_probe.err
echo "== GAN generator step with the adaptive-weight pin"; date
timeout 600 python -m pytest tests/test_gan_gpu.py -m gpu -q -p no:cacheprovider -k "generator_gan_step_matches_reference or forward_surface or pixel_row" > gpurun_out/r5c_gan.log 2>&1; echo "gan rc=$?"; tail -15 gpurun_out/r5c_gan.log
grep adaptive gpurun_out/parity.jsonl
date
