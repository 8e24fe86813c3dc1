#!/bin/bash
# A/B of one environment switch on a bench workload, each setting in a process of its own,
# interleaved twice:  tools/env_ab.sh WORKLOAD VAR v1 v2 ...   ->  gpurun_out/env_ab_<VAR>.log
set -u
wl=$1; var=$2; shift 2
out=gpurun_out/env_ab_${var}_${wl}.log
: > $out
for rep in 1 2; do
    for v in "$@"; do
        echo -n "$var=$v  " >> $out
        env $var=$v timeout -k 10 150 python bench.py --workload $wl --no-also --no-cpu-baseline \
            --steps 40 --warmup 10 2>/dev/null \
            | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"])' >> $out || exit 1
    done
done
cat $out
