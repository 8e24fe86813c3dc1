set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PE_RESNET_DOWN_STREAM=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sp -o p -- python $R/bench.py --workload infer --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_sp.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof_sp/**/p_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:8]:
    print(r['Name'][:100], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['TotalDurationNs'])/1e6,2))
PY
