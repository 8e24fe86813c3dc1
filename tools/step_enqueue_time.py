#!/usr/bin/env python3
"""Host enqueue time of one step of a bench workload next to its wall time: if the two are
close the step is launch-bound.      python tools/step_enqueue_time.py <workload> [steps]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def main():
    wl = bench.WORKLOADS[sys.argv[1]]()
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    wl.setup(dev)
    for i in range(3):
        wl.step(i)
    torch.cuda.synchronize()
    enq = 0.0
    t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        wl.step(i)
        enq += time.perf_counter() - a
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"{wl.name}: host enqueue {enq / n * 1e3:.2f} ms per step, wall {tot / n * 1e3:.2f} ms per step")
    x = torch.empty(1 << 28, device=dev)
    torch.cuda.synchronize()
    for _ in range(60):
        x.add_(1.0)
    a = time.perf_counter()
    for i in range(n):
        wl.step(i)
    enq2 = time.perf_counter() - a
    torch.cuda.synchronize()
    print(f"host enqueue with the GPU kept busy: {enq2 / n * 1e3:.2f} ms per step")


if __name__ == "__main__":
    main()
