#!/bin/bash
# hardware counters of the two Schwarz apply kernels (separate passes; kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/pmc_list.txt 2>&1
for kind in ${KINDS:-6 7}; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_apply_a$kind -o a -- python3 $R/tools/apply_once.py 214 $kind 6 > $R/gpurun_out/pmc_apply_a$kind.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/pmc_apply_b$kind -o b -- python3 $R/tools/apply_once.py 214 $kind 6 > $R/gpurun_out/pmc_apply_b$kind.log 2>&1
  rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -d $R/gpurun_out/pmc_apply_c$kind -o c -- python3 $R/tools/apply_once.py 214 $kind 6 > $R/gpurun_out/pmc_apply_c$kind.log 2>&1
done
ls $R/gpurun_out/pmc_apply_*
