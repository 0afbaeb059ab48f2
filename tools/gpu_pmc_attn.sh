#!/bin/bash
# SQ counters of the n = 576 attention kernels (tools/attn_bias_bench.py runs every variant once per graph capture)
set -u
export TMPDIR=/tmp
out=gpurun_out/pmc_attn
rm -rf $out ${out}_b; mkdir -p $out ${out}_b
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
  --output-format csv -d $out -o p -- python tools/attn_bias_bench.py > $out/log.txt 2>&1
python tools/pmc_sq.py $out | grep -A9 "attn_fwd_lds_kernel"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES \
  --output-format csv -d ${out}_b -o p -- python tools/attn_bias_bench.py > ${out}_b/log.txt 2>&1
python tools/pmc_sq.py ${out}_b | grep -A9 "attn_fwd_lds_kernel"
