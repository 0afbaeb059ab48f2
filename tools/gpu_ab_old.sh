#!/bin/bash
# same-box A/B of the whole bench: the tree of an older commit (unpacked + built under tools/_bin/old) against the working tree.
# Prepare HERE (the GPU box has no .git), then run this script through gpurun:
#   rm -rf tools/_bin/old && mkdir -p tools/_bin/old && git archive <commit> | tar -x -C tools/_bin/old && \
#     rm -rf tools/_bin/old/tests/golden tools/_bin/old/profiles && (cd tools/_bin/old && python -m phenaki_pytorch_amd.build)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do
  for which in old new; do
    if [ $which = old ]; then dir=tools/_bin/old; else dir=.; fi
    (cd $dir && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-kernels --no-graph --groups 7 2>/dev/null) > gpurun_out/ab_$which.json
    python - <<PY
import json
d=json.load(open('gpurun_out/ab_$which.json')); print('$which', 'encode ms', round(d['ms_per_step'],4), 'decode', round(d['decode'].get('ms_per_step'),4), 'sample', round(d['sample']['seconds_per_sample_call'],5), 'cfg3', round(d['sample_cfg3']['seconds_per_sample_call'],5), 'mv', round(d['make_video']['wall_clock_s'],4))
PY
  done
done
