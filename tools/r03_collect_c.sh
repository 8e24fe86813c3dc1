#!/bin/bash
# Round-3 evidence, part C (GPU box, repository root), after the fused layer1 launches: bench
# lines + kernel statistics of the two inference workloads they changed, the per-layer table,
# the HBM traffic passes of the forward, and the MFMA-busy counters of the 3x3 shapes.
set -e -o pipefail
bash tools/refresh_profiles.sh "infer hrnet"
python tools/infer_layer_table.py gpurun_out/refresh/infer_kernel_trace.csv 256 > gpurun_out/refresh/infer_layer_table.txt
tail -4 gpurun_out/refresh/infer_layer_table.txt
PE_RESNET_DOWN_STREAM=0 bash tools/collect_traffic.sh infer
bash tools/pmc_mfma.sh gpurun_out/pmc_mfma_r3 256,16,12,256,256,3 256,32,24,128,128,3 256,64,48,64,64,3 256,8,6,512,512,3 256,32,24,256,256,3 256,64,48,64,256,1
for d in gpurun_out/pmc_mfma_r3/*/; do n=$(basename $d); find $d -name "p_counter_collection.csv" -exec cp {} gpurun_out/pmc_mfma_r3/$n.csv \; ; done
python tools/mfma_pmc_table.py gpurun_out/pmc_mfma_r3
