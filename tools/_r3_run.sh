set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, torch, subprocess, sys
PY
for d in 0 25 50 75 100 150 200 300; do
  echo "== STAGGER=$d"
  PE_SP_STAGGER=$d timeout -k 10 120 python - <<'PY'
import torch, time
from pose_estimation_amd import nn as pnn, synthetic
B=256
xs = pnn.preprocess_u8(torch.from_numpy(synthetic.images_u8(B,256,192,seed=1)).cuda())
w = torch.randn(64,4,64).to(torch.bfloat16).cuda()
sc = torch.rand(64).cuda()+0.5; sh = torch.randn(64).cuda()
out = torch.empty(B,64,48,64,dtype=torch.bfloat16,device='cuda')
for _ in range(5): pnn.stem7x7s2_pool(xs,w,sc,sh,out=out)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): pnn.stem7x7s2_pool(xs,w,sc,sh,out=out)
e1.record(); torch.cuda.synchronize()
print("us per launch", e0.elapsed_time(e1)*1e3/50)
PY
done
