#!/bin/bash
# A/B of the depth-first front of the ResNet-50 forward (PE_RESNET_CHUNK / PE_RESNET_CHUNK_BLOCKS):
# each setting in a process of its own, interleaved twice.  Output: gpurun_out/chunk_ab.log
set -u
out=gpurun_out/chunk_ab.log
: > $out
run() {  # chunk blocks
    echo "== chunk $1 blocks $2" >> $out
    PE_RESNET_CHUNK=$1 PE_RESNET_CHUNK_BLOCKS=$2 timeout -k 10 120 python bench.py --workload infer \
        --no-also --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null \
        | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"])' >> $out || exit 1
}
for rep in 1 2; do
    run 0 3 || exit 1
    for cfg in "$@"; do
        run ${cfg%:*} ${cfg#*:} || exit 1
    done
done
cat $out
