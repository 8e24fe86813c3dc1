#!/bin/bash
# A/B of the training workloads under an environment switch, interleaved on one box.
# usage: tools/ab_train.sh VAR "workload ..." [rounds]   (VAR=0 against VAR unset)
VAR=$1; WLS=${2:-"train hrnet_train mobilenet_train"}; R=${3:-2}
for r in $(seq $R); do
  for WL in $WLS; do
    for V in unset 0; do
      if [ $V = unset ]; then
        OUT=$(timeout -k 10 170 python3 bench.py --workload $WL --steps 30 --warmup 8 --no-cpu-baseline) || exit 1
      else
        OUT=$(env $VAR=0 timeout -k 10 170 python3 bench.py --workload $WL --steps 30 --warmup 8 --no-cpu-baseline) || exit 1
      fi
      echo "$WL $VAR=$V $(echo "$OUT" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.3f ms/step" % d["ms_per_step"])')"
    done
  done
done
