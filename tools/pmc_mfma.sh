#!/bin/bash
# One hardware-counter pass (MFMA busy cycles next to kernel cycles) over a few conv shapes,
# its own short run per shape.   usage: tools/pmc_mfma.sh <outdir> <shape> [shape ...]
set -u
OUT=$1; shift
R=$(pwd)
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
P="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for S in "$@"; do
    T=$(echo $S | tr ',' '_')
    timeout -s KILL 90 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/$OUT/$T -o p -- python3 $R/tools/conv_bench.py $S --iters 3 > $R/$OUT/$T.log 2>&1 || { echo "pass $S failed"; tail -3 $R/$OUT/$T.log; exit 1; }
    find $R/$OUT/$T -name "*.db" -delete
done
echo done
