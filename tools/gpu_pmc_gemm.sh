#!/bin/bash
# SQ counters of one GEMM shape (separate pass from any trace): bash tools/gpu_pmc_gemm.sh M N K variant mode
set -u
export TMPDIR=/tmp
out=gpurun_out/pmc_gemm_$1_$2_$3_$4_${5:-plain}
rm -rf $out ${out}_b; mkdir -p $out ${out}_b
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
  --output-format csv -d $out -o p -- python tools/one_gemm.py $1 $2 $3 $4 10 ${5:-} > $out/log.txt 2>&1
python tools/pmc_sq.py $out
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE \
  --output-format csv -d ${out}_b -o p -- python tools/one_gemm.py $1 $2 $3 $4 10 ${5:-} > ${out}_b/log.txt 2>&1
python tools/pmc_sq.py ${out}_b
