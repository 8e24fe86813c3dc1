#!/bin/bash
# Runs on the GPU box: bench lines and rocprofv3 kernel statistics for the main workloads.
# Results land under gpurun_out/refresh/; copy what should be kept into profiles/.
#   tools/refresh_profiles.sh "decode infer train mobilenet"
set -e -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
for wl in ${1:-decode infer train mobilenet}; do
    echo "== $wl"
    timeout -k 10 240 python3 bench.py --workload "$wl" > "$OUT/bench_$wl.json"
    tail -c 600 "$OUT/bench_$wl.json"
    echo
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$wl" -o "$wl" -- \
        python3 bench.py --workload "$wl" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/prof_$wl.log" 2>&1
    find "$OUT/prof_$wl" -name "*kernel_stats.csv" -exec cp {} "$OUT/${wl}_kernel_stats.csv" \;
    find "$OUT/prof_$wl" -name "*_kernel_trace.csv" -delete
    find "$OUT/prof_$wl" -name "*.db" -delete
done
