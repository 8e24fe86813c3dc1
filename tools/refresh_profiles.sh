#!/bin/bash
# Runs on the GPU box: one bench line and one rocprofv3 kernel-statistics summary per workload,
# each workload alone in its trace (--no-also), so a CSV's rows belong to that workload only.
# Results land under gpurun_out/refresh/; copy what should be kept into profiles/.
#   tools/refresh_profiles.sh "decode infer train mobilenet"
# The inference trace runs with PE_RESNET_DOWN_STREAM=0: every launch on one stream, so the
# per-kernel durations add up to the step instead of overlapping.
set -e -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
for wl in ${1:-decode infer train mobilenet}; do
    echo "== $wl"
    timeout -k 10 240 python3 bench.py --workload "$wl" --no-also > "$OUT/bench_$wl.json"
    tail -c 700 "$OUT/bench_$wl.json"
    echo
    if [ "$wl" = infer ]; then export PE_RESNET_DOWN_STREAM=0; else unset PE_RESNET_DOWN_STREAM; fi
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$wl" -o "$wl" -- \
        python3 "$ROOT/bench.py" --workload "$wl" --steps 10 --warmup 3 --no-cpu-baseline --no-also > "$OUT/prof_$wl.log" 2>&1)
    find "$OUT/prof_$wl" -name "*kernel_stats.csv" -exec cp {} "$OUT/${wl}_kernel_stats.csv" \;
    find "$OUT/prof_$wl" -name "*_kernel_trace.csv" -exec cp {} "$OUT/${wl}_kernel_trace.csv" \;
    find "$OUT/prof_$wl" -name "*.db" -delete
done
