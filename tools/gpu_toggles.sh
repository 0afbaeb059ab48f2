#!/bin/bash
# same-box effect of each round-2 switch on every bench leg (working tree; one switch turned off at a time), old tree for reference
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # name dir env...
  name=$1; dir=$2; shift 2
  (cd $dir && env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-mode --no-kernels --no-graph --groups 7 2>/dev/null) > gpurun_out/tg_$name.json
  python - <<PY
import json
d=json.load(open('gpurun_out/tg_$name.json')); print(f"{'$name':22s}", 'encode ms', round(d['ms_per_step'],4), 'decode', round(d['decode'].get('ms_per_step'),4), 'sample', round(d['sample']['seconds_per_sample_call'],5), 'cfg3', round(d['sample_cfg3']['seconds_per_sample_call'],5), 'mv', round(d['make_video']['wall_clock_s'],4))
PY
}
[ -d tools/_bin/old ] && run old tools/_bin/old X=1
run new . X=1
run ff_fold_off . PK_LN_FOLD_FF=0
run shared_prefix_off . PK_CFG_SHARED_PREFIX=0
run attn_fixed_off . PK_ATTN_FIXED=0
run attn_qf1 . PK_ATTN_LDS_QF=1
[ -d tools/_bin/old ] && run old_again tools/_bin/old X=1
run new_again . X=1
