#!/usr/bin/env python3
"""cProfile of the host side of training steps (where the enqueue time goes)."""
import cProfile
import pstats
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from pose_estimation_amd import model, synthetic  # noqa: E402
from pose_estimation_amd.training import TrainablePoseResNet  # noqa: E402

dev = torch.device("cuda", 0)
net = TrainablePoseResNet(dev)
net.load_state(model.init_state(net.specs, seed=3))
img = torch.from_numpy(synthetic.images_u8(64, 256, 192, seed=1)).to(dev)
j = torch.from_numpy(synthetic.joints(64, 17, 256, 192, seed=2)).to(dev)
for _ in range(5):
    net.train_step(img, j, None)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    net.train_step(img, j, None)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
