#!/bin/bash
# One rocprofv3 counter pass over a bench workload: tools/pmc_pass.sh <workload> <tag> <counters...>
# CSV under gpurun_out/pmc_<workload>/<tag>/ (a pass holds at most 8 SQ_, 4 TCC_, 4 TCP_, 2 TA_).
set -u
R=$(pwd); WL=$1; TAG=$2; shift 2
D=$R/gpurun_out/pmc_$WL/$TAG; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $D -o p -- python3 $R/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-also > $D/run.log 2>&1 || { echo "$WL $TAG failed"; tail -3 $D/run.log; exit 1; }
python3 - "$D/p_counter_collection.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
