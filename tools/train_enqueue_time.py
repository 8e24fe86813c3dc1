#!/usr/bin/env python3
"""How long the host needs to enqueue one training step, next to how long the GPU needs to run
it: if the two are close the step is launch-bound.

    python tools/train_enqueue_time.py [batch]
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from pose_estimation_amd import model, synthetic  # noqa: E402
from pose_estimation_amd.training import TrainablePoseResNet  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda", 0)
    net = TrainablePoseResNet(dev)
    net.load_state(model.init_state(net.specs, seed=3))
    img = torch.from_numpy(synthetic.images_u8(B, 256, 192, seed=1)).to(dev)
    j = torch.from_numpy(synthetic.joints(B, 17, 256, 192, seed=2)).to(dev)
    for _ in range(5):
        net.train_step(img, j, None)
    torch.cuda.synchronize()
    n = 20
    enq = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter()
        net.train_step(img, j, None)
        enq += time.perf_counter() - a
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"batch {B}: host enqueue {enq / n * 1e3:.2f} ms per step, wall {tot / n * 1e3:.2f} ms per step")
    # the same with the host far ahead: enqueue while the GPU is still busy with a long kernel
    x = torch.empty(1 << 28, device=dev)
    torch.cuda.synchronize()
    for _ in range(40):
        x.add_(1.0)
    a = time.perf_counter()
    for _ in range(n):
        net.train_step(img, j, None)
    enq2 = time.perf_counter() - a
    torch.cuda.synchronize()
    print(f"host enqueue with the GPU kept busy: {enq2 / n * 1e3:.2f} ms per step")


if __name__ == "__main__":
    main()
