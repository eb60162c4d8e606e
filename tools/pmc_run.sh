#!/bin/bash
# Counter passes for one kernel of tools/pmc_kernels.py on the GPU box (each --pmc set in its own run, kernel-trace only):
#   bash tools/pmc_run.sh <which> <out_prefix> [Bc] [L]     ->  gpurun_out/<out_prefix>_{a,b,c,d}/ ; reduce with tools/pmc_reduce.py
set -u
W=$1; P=$2; BC=${3:-10}; LEN=${4:-352}
export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"
for pass in a b c d; do
  case $pass in a) C="$A";; b) C="$B";; c) C="FETCH_SIZE";; d) C="WRITE_SIZE";; esac
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/${P}_$pass -- python tools/pmc_kernels.py $W $BC $LEN > gpurun_out/${P}_$pass.log 2>&1
done
