#!/bin/bash
# Forward implicit GEMM at the training batch (64): automatic tile choice against forced ring
# configurations, per shape.   tools/igemm_small_sweep.sh
SHAPES="64,8,6,512,512,3 64,16,12,256,256,3 64,32,24,128,128,3 64,8,6,2048,512,1 64,8,6,512,2048,1 64,16,12,1024,256,1 64,16,12,256,1024,1 64,32,24,512,128,1 64,32,24,128,512,1"
for cfg in auto "64,64,2,3,0,2,2" "64,64,2,4,0,2,2" "64,128,2,3,0,2,2" "128,128,2,2,0,2,2" "128,128,2,3,0,1,2"; do
  echo "== $cfg"
  if [ "$cfg" = auto ]; then unset PE_IGEMM_CFG; else export PE_IGEMM_CFG=$cfg; fi
  timeout -k 10 120 python tools/conv_bench.py $SHAPES 2>&1 | grep -v amdgpu || exit 1
done
