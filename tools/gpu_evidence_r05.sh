#!/bin/bash
# round-5 evidence on ONE box: full GPU tests (parity records), smoke, the bench exactly as the driver runs it, rocprofv3 kernel stats and the PMC passes
set -u
cd "$(dirname "$0")/.."
bash tools/gpu_evidence.sh
cp gpurun_out/parity.jsonl gpurun_out/parity_r05_full.jsonl 2>/dev/null
PK_ROUND=r05 bash tools/gpu_profile.sh
