#!/bin/bash
# Round-3 evidence, part A (run on the GPU box from the repository root): conv parity, then one
# bench line + rocprofv3 kernel statistics per workload, then the per-layer inference table.
set -e -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_conv.py -q -x --timeout 200 --timeout-method thread 2>&1 | tail -2
bash tools/refresh_profiles.sh "decode infer train hrnet_train hrnet mobilenet mobilenet_train"
python tools/infer_layer_table.py gpurun_out/refresh/infer_kernel_trace.csv 256 > gpurun_out/refresh/infer_layer_table.txt
tail -5 gpurun_out/refresh/infer_layer_table.txt
python tools/train_kernel_table.py gpurun_out/refresh/train_kernel_trace.csv > gpurun_out/refresh/train_kernel_table.txt 2>&1 || true
head -12 gpurun_out/refresh/train_kernel_table.txt
