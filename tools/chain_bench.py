#!/usr/bin/env python3
"""Times a layer1 bottleneck tail (3x3 64->64, 1x1 64->256 + residual + ReLU, 1x1 256->64) at
batch 256, 64x48: every layer as a launch of its own against the chained launch, by prefix."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pose_estimation_amd import nn as pnn  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
B, H, W = 256, 64, 48
nbuf = 3
xs = [torch.randn((B, H, W, 64), generator=g).to(torch.bfloat16).to(dev) for _ in range(nbuf)]
res = [torch.randn((B, H, W, 256), generator=g).to(torch.bfloat16).to(dev) for _ in range(nbuf)]


def mk(co, ci, k):
    return {"w": pnn.pack_conv_weight(torch.randn((co, ci, k, k), generator=g) * 0.05).to(dev),
            "scale": torch.ones(co, device=dev), "shift": torch.zeros(co, device=dev),
            "pad": k // 2, "relu": True}


st = [mk(64, 64, 3), mk(256, 64, 1), mk(64, 256, 1)]
outs = [torch.empty((B, H, W, c), dtype=torch.bfloat16, device=dev) for c in (64, 256, 64)]
for s, o in zip(st, outs):
    s["out"] = o


def timed(fn, iters=30):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def separate(n):
    def fn(i):
        cur = xs[i % nbuf]
        for k in range(n):
            s = st[k]
            cur = pnn.conv2d(cur, s["w"], s["scale"], s["shift"], res[i % nbuf] if k == 1 else None,
                             1, s["pad"], True, out=s["out"])
    return fn


def chained(n):
    def fn(i):
        st[1]["residual"] = res[i % nbuf]
        pnn.conv2d_chain(xs[i % nbuf], st[:n])
    return fn


for n in (1, 2, 3):
    print(f"{n} stage(s): separate {timed(separate(n)):7.1f} us   chained {timed(chained(n)):7.1f} us")
