#!/usr/bin/env python3
"""What the host pays to enqueue one training step, eager against the recorded call list: the
workload is run at a small batch, where the GPU finishes each launch faster than the host can
issue the next one, so the wall time per step IS the host's enqueue cost.
    python tools/replay_host_cost.py <workload> [batch] [steps]"""
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def main():
    name = sys.argv[1]
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for mode in ("0", "1"):
        os.environ["PE_HOST_REPLAY"] = mode
        wl = bench.WORKLOADS[name]()
        wl.batch = batch
        wl.setup(dev)
        for i in range(3):
            wl.step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            wl.step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{wl.name} batch {batch} PE_HOST_REPLAY={mode}: enqueue {(t1 - t0) / n * 1e3:.2f} ms, "
              f"wall {(t2 - t0) / n * 1e3:.2f} ms per step", flush=True)
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
