#!/bin/bash
# Hardware-counter passes over one conv shape (each pass is its own short run).
# usage: tools/pmc_conv.sh <shape> <outdir>
set -u
SHAPE=$1
OUT=$2
R=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
P3="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_CYCLES"
i=1
for P in "$P1" "$P2" "$P3"; do
    timeout -s KILL 90 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/$OUT/p$i -o p -- python3 $R/tools/conv_bench.py $SHAPE --iters 3 > $R/$OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/$OUT/p$i.log; exit 1; }
    i=$((i+1))
done
echo done
