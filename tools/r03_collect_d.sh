#!/bin/bash
# Round-3 evidence, part D (GPU box, repository root), after the next block's conv1 joined the
# fused conv2 + conv3 launch: bench lines + kernel statistics of the two inference workloads it
# changed, the per-layer table and the HBM traffic passes of the forward.
set -e -o pipefail
bash tools/refresh_profiles.sh "infer hrnet"
python tools/infer_layer_table.py gpurun_out/refresh/infer_kernel_trace.csv 256 > gpurun_out/refresh/infer_layer_table.txt
tail -4 gpurun_out/refresh/infer_layer_table.txt
PE_RESNET_DOWN_STREAM=0 bash tools/collect_traffic.sh infer
