#!/bin/bash
# Forces each implicit-GEMM tile configuration in turn: conv parity tests, then layer timings.
# usage: tools/igemm_sweep.sh "cfg1 cfg2 ..." ["shape1 shape2 ..."]
#   cfg = BM,BN,WM,DB,PF,MINB; "default" = launcher's choice
set -u
SHAPES=${2:-"256,16,12,256,256,3 256,32,24,128,128,3 256,64,48,64,64,3 256,8,6,512,512,3 256,64,48,64,256,1 256,64,48,256,64,1 256,16,12,1024,256,1"}
for cfg in $1; do
    echo "== cfg $cfg"
    if [ "$cfg" != default ]; then export PE_IGEMM_CFG=$cfg; fi
    timeout -k 10 300 python -u -m pytest tests/test_gpu_conv.py -m gpu -q -x --timeout 120 --timeout-method thread 2>&1 | tail -1 || exit 1
    timeout -k 10 200 python tools/conv_bench.py $SHAPES || exit 1
done
