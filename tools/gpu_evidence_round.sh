#!/bin/bash
# round evidence on ONE box: full GPU tests (parity records), smoke, the bench exactly as the driver runs it, the split-bf16 first-frame split-K A/B,
# rocprofv3 kernel stats (encode leg, all legs, training legs) and the PMC traffic passes
set -u
cd "$(dirname "$0")/.."
bash tools/gpu_evidence.sh
echo "== bf16x3 first-frame patch-embedding split-K A/B"; date
for v in 8 1 8 1; do PK_PE_SPLITK_FIRST=$v python bench.py --dtype bf16x3 --encode-only --groups 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PK_PE_SPLITK_FIRST=$v', round(d['value']), round(d['ms_per_step'],4))"; done
bash tools/gpu_profile.sh
