#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the encode leg and of the whole bench, PMC traffic passes (separate runs).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${PK_ROUND:-r04}
P=gpurun_out/prof_$R
rm -rf $P; mkdir -p $P
echo "== kernel trace: encode leg"; date
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/enc -o enc -- python bench.py --encode-only --groups 3 --steps 20 --warmup 3 > $P/enc.json 2> $P/enc.err; echo rc=$?
echo "== kernel trace: whole bench (no cpu / parity mode)"; date
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/all -o all -- python bench.py --no-cpu --no-parity-mode --no-kernels --groups 3 --legs decode,sample,sample_cfg3,make_video,objective > $P/all.json 2> $P/all.err; echo rc=$?
echo "== kernel trace: training legs (Phenaki step in bf16x3 + bf16, tokenizer step, tokenizer GAN step)"; date
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/train -o train -- python bench.py --no-cpu --no-parity-mode --no-kernels --groups 2 --legs train_step,cvivit_train_step,cvivit_gan_step > $P/train.json 2> $P/train.err; echo rc=$?
echo "== steady-state launch census of the Phenaki training step (bf16x3; tools/train_census.py)"; date
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$P/census -o c -- python $OLDPWD/tools/train_census.py bf16x3 6 phenaki > $OLDPWD/$P/census.log 2>&1 ); echo rc=$?
python tools/train_census.py --summary $P/census 6 > $P/train_step_census_$R.txt; head -12 $P/train_step_census_$R.txt
echo "== PMC FETCH_SIZE"; date
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -o p -- python bench.py --encode-only --no-graph --no-kernels --groups 1 --steps 3 --warmup 1 > /dev/null 2> $P/pmc_fetch.err; echo rc=$?
echo "== PMC WRITE_SIZE"; date
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -o p -- python bench.py --encode-only --no-graph --no-kernels --groups 1 --steps 3 --warmup 1 > /dev/null 2> $P/pmc_write.err; echo rc=$?
# per-leg sections (no null roofline.traffic): the bf16 sampling leg, and the parity-grade bf16x3 mode's encode / sampling legs
pmc_pass() {   # name counter args...
  local name=$1 ctr=$2; shift 2
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $P/$name -o p -- python bench.py --no-graph --no-kernels --no-cpu --no-parity-mode --groups 1 --steps 2 --warmup 1 "$@" > /dev/null 2> $P/$name.err; echo "$name rc=$?"
}
echo "== PMC sample leg (bf16)"; date
pmc_pass pmc_s_fetch FETCH_SIZE --legs sample
pmc_pass pmc_s_write WRITE_SIZE --legs sample
echo "== PMC bf16x3 encode / sample"; date
pmc_pass pmc_x3e_fetch FETCH_SIZE --dtype bf16x3 --encode-only
pmc_pass pmc_x3e_write WRITE_SIZE --dtype bf16x3 --encode-only
pmc_pass pmc_x3s_fetch FETCH_SIZE --dtype bf16x3 --legs sample
pmc_pass pmc_x3s_write WRITE_SIZE --dtype bf16x3 --legs sample
python tools/pmc_traffic.py $P/pmc_fetch $P/pmc_write $P/pmc_traffic_$R.json sample $P/pmc_s_fetch $P/pmc_s_write bf16x3_encode $P/pmc_x3e_fetch $P/pmc_x3e_write bf16x3_sample $P/pmc_x3s_fetch $P/pmc_x3s_write | head -70
find $P -name "*kernel_stats.csv" | head
# keep the merge-back small: drop the per-dispatch traces, keep the stats
find $P -name "*kernel_trace.csv" -delete; find $P -name "*counter_collection.csv" -delete; find $P -name "*.db" -delete
du -sh $P
echo "== done"; date
