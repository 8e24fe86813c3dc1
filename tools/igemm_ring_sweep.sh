#!/bin/bash
# Times the forward implicit GEMM of representative ResNet-50 pose layers (batch 256) under each
# forced tile / pipeline configuration ("BM,BN,WM,PIPE,PF,MINB,WN"; "auto" = the library's choice).
# usage: tools/igemm_ring_sweep.sh [config ...]      output: one block per configuration
set -u
SHAPES="256,32,24,128,128,3 256,32,24,256,256,3 256,16,12,256,256,3 256,16,12,1024,256,1 256,16,12,256,1024,1 256,8,6,512,512,3 256,8,6,2048,512,1 256,64,48,64,64,3 256,32,24,512,128,1 256,64,48,256,64,1"
CFGS=${@:-"auto 256,128,4,3,0,1,2 256,128,4,2,0,1,2 256,256,4,2,0,1,2 256,256,2,2,0,1,4 128,256,2,3,0,1,4 128,128,2,3,0,1,2 128,128,2,4,0,1,2 128,128,2,2,0,2,2 128,128,4,3,0,1,2 128,128,4,4,0,1,2 128,128,2,4,0,1,4 128,64,4,4,0,1,2 128,64,2,3,0,2,2 256,64,4,3,0,1,2"}
for C in $CFGS; do
    echo "== $C"
    if [ "$C" = auto ]; then
        timeout -k 10 120 python3 tools/conv_bench.py $SHAPES --iters 30 || exit 1
    else
        PE_IGEMM_CFG=$C timeout -k 10 120 python3 tools/conv_bench.py $SHAPES --iters 30 || exit 1
    fi
done
