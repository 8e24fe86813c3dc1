#!/bin/bash
# Round-3 evidence, part B (GPU box, repository root): HBM traffic counter passes of the
# inference forward and the two training steps (fold them with tools/infer_traffic_table.py and
# tools/step_traffic.py afterwards; they only parse the CSVs), then the MFMA-busy counters of
# the 3x3 shapes.
set -e
if [ "${1:-all}" != pmc ]; then
  PE_RESNET_DOWN_STREAM=0 bash tools/collect_traffic.sh infer
  bash tools/collect_traffic.sh train hrnet_train
fi
bash tools/pmc_mfma.sh gpurun_out/pmc_mfma_r3 256,16,12,256,256,3 256,32,24,128,128,3 256,64,48,64,64,3 256,8,6,512,512,3 256,32,24,256,256,3 256,64,48,64,256,1
for d in gpurun_out/pmc_mfma_r3/*/; do n=$(basename $d); find $d -name "p_counter_collection.csv" -exec cp {} gpurun_out/pmc_mfma_r3/$n.csv \; ; done
python tools/mfma_pmc_table.py gpurun_out/pmc_mfma_r3
