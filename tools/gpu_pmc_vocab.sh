#!/bin/bash
# SQ counters of the vocabulary head at 4608 x 65536 x 512 (bf16), resident (8- and 12-wave builds) vs tiled kernel; counters in their own passes
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for r in 1 2 0; do
  out=gpurun_out/pmc_vocab_res$r
  rm -rf $out ${out}_b; mkdir -p $out ${out}_b
  PK_VOCAB_RESIDENT=$r rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
    --output-format csv -d $out -o p -- python tools/vocab_bench.py 4608 > $out/log.txt 2>&1
  echo "== PK_VOCAB_RESIDENT=$r"; python tools/pmc_sq.py $out | grep -A9 "vocab_" | head -24
  PK_VOCAB_RESIDENT=$r rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE \
    --output-format csv -d ${out}_b -o p -- python tools/vocab_bench.py 4608 > ${out}_b/log.txt 2>&1
  python tools/pmc_sq.py ${out}_b | grep -A9 "vocab_" | head -24
  find $out ${out}_b -name "*.db" -delete; find $out ${out}_b -name "*kernel_trace.csv" -delete
done
