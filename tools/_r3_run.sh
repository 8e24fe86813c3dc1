set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -x -q --timeout 200 --timeout-method thread -k "resident or resnet or predict or small" > gpurun_out/r3_pc_tests.log 2>&1 || { tail -40 gpurun_out/r3_pc_tests.log; exit 1; }
tail -3 gpurun_out/r3_pc_tests.log
for f in 0 1 0 1; do
  echo "== PATCH3X3=$f"
  PE_RESNET_PATCH3X3=$f timeout -k 10 120 python bench.py --workload infer --steps 30 --warmup 5 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
