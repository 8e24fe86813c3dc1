#!/usr/bin/env python3
"""Does the ResNet-50 training step slow down when the process has put work on other streams
first (as the collective's stream and the inference leg's side stream do in an N > 1 bench run)?
HIP gives a process four hardware queues; streams beyond that share them.

    python tools/queue_share_check.py <number of streams used before the training streams> [workload]
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

n = int(sys.argv[1])
name = sys.argv[2] if len(sys.argv) > 2 else "train"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
keep = []
for _ in range(n):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        keep.append(torch.zeros(1024, device=dev) + 1)      # a launch: the stream gets its queue
torch.cuda.synchronize()
wl = bench.WORKLOADS[name]()
wl.setup(dev)
for i in range(8):
    wl.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(30):
    wl.step(i)
torch.cuda.synchronize()
print(f"{name}: {n} streams used first: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step")
