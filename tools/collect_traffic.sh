#!/bin/bash
# HBM traffic of a bench workload's dominant kernel: two counter passes (FETCH_SIZE, WRITE_SIZE),
# each its own short run, CSV output under gpurun_out/traffic/<workload>/{fetch,write}.
# usage: tools/collect_traffic.sh <workload> [<workload> ...]
set -u
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for WL in "$@"; do
    for C in FETCH_SIZE WRITE_SIZE; do
        D=$R/gpurun_out/traffic/$WL/$(echo $C | tr A-Z a-z | cut -d_ -f1)
        mkdir -p $D
        timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- python3 $R/bench.py --workload $WL --steps 20 --warmup 2 --no-cpu-baseline > $D/run.log 2>&1 || { echo "$WL $C failed"; tail -3 $D/run.log; exit 1; }
    done
    echo "$WL done"
done
