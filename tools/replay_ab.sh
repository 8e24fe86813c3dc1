#!/bin/bash
# Eager enqueue against the recorded call list, per training workload: host enqueue time and
# wall time per step (tools/step_enqueue_time.py), same box, back to back.
#   tools/replay_ab.sh "train hrnet_train mobilenet_train" [steps]
set -u
for wl in $1; do
  for r in 0 1; do
    echo "== $wl PE_HOST_REPLAY=$r"
    PE_HOST_REPLAY=$r timeout -k 10 200 python tools/step_enqueue_time.py $wl ${2:-20} 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
