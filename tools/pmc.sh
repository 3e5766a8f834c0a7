#!/bin/bash
# PMC passes (each its own rocprofv3 run, --kernel-trace only) for the stage kernel.  Usage: tools/pmc.sh <outdir> <cmd...>
set -u
OUT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
