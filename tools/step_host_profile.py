#!/usr/bin/env python3
"""cProfile of the host side of one bench workload's step.
    python tools/step_host_profile.py <workload> [steps]"""
import cProfile
import pstats
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1]]()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
wl.setup(dev)
for i in range(3):
    wl.step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    wl.step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
