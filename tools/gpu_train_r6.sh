#!/bin/bash
# round-6 training-step legs, one sub-command per gpurun call:  bash tools/gpu_train_r6.sh <leg>
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
leg=${1:-test}; shift || true
case $leg in
test)   # the training parity tests (blocks, whole step vs the reference's autograd, tokenizer, GAN step)
  timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_train_cvivit_gpu.py tests/test_gan_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -15
  ;;
ab)     # same-box A/B against an export of the previous tree under _ab_old/ (alternating)
  for i in 1 2; do
    for m in ${1:-bf16x3 bf16}; do
      [ -d _ab_old ] && timeout 300 python tools/train_time.py --root _ab_old --mode $m --loss 2>&1 | tail -4
      timeout 300 python tools/train_time.py --mode $m --loss 2>&1 | tail -4
    done
  done
  ;;
env)    # same-box A/B of environment switches of THIS tree:  env "PK_X=0" "PK_X=1" ...
  for i in 1 2; do
    for e in "$@"; do
      echo "== $e"; env $e timeout 300 python tools/train_time.py --mode bf16x3 --loss 2>&1 | tail -2
    done
  done
  ;;
census) # launches and kernel time per step
  R=$GRAFT_REPO_ROOT
  rm -rf $R/gpurun_out/census; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/census -o c -- python $R/tools/train_census.py ${1:-bf16x3} 6 ${2:-phenaki} > $R/gpurun_out/census_run.log 2>&1
  cd $R && tail -2 gpurun_out/census_run.log && python tools/train_census.py --summary gpurun_out/census 6 | tee gpurun_out/train_step_census_r06_${2:-phenaki}.txt | head -${3:-90}
  ;;
esac
