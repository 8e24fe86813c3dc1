#!/usr/bin/env python3
"""Times the one-launch batch-norm forward (finalize folded into the element-wise pass) against
the separate finalize + apply launches, HIP events around a loop.
    python tools/bn_fused_bench.py [B,H,W,C ...]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pose_estimation_amd import nn as pnn  # noqa: E402


def bench(shape, iters=200):
    B, H, W, C = shape
    dev = torch.device("cuda", 0)
    M = B * H * W
    z = torch.randn(shape, device=dev).to(torch.bfloat16)
    gamma = torch.ones(C, device=dev)
    beta = torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ws = pnn.bn_workspace(M, C, dev)
    rows = torch.empty(4, C, device=dev)
    sync = pnn.bn_sync(dev)
    y = torch.empty_like(z)
    pnn.bn_stats(z, gamma, beta, 1e-5, 0.1, rm, rv, ws, out=rows)   # leaves partial rows in ws
    nrows = min(1024, (M + 63) // 64)

    def separate():
        _, _, sc, sh = pnn.bn_finalize(ws, nrows, M, C, gamma, beta, 1e-5, 0.1, rm, rv, out=rows)
        pnn.bn_apply(z, sc, sh, None, 1, out=y)

    def fused():
        pnn.bn_finalize_apply(ws, nrows, z, gamma, beta, 1e-5, 0.1, rm, rv, rows, sync, None, 1, y)

    res = []
    for fn in (separate, fused):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / iters)
    mb = 2 * z.numel() * 2 / 1e6
    print(f"{shape}: {mb:7.1f} MB  separate {res[0]:7.1f} us  fused {res[1]:7.1f} us  "
          f"timeouts {int(sync[512])}", flush=True)


if __name__ == "__main__":
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
        (64, 64, 48, 64), (64, 64, 48, 256), (64, 32, 24, 128), (64, 16, 12, 1024),
        (64, 8, 6, 2048), (64, 96, 72, 32), (64, 24, 18, 128)]
    for s in shapes:
        bench(s)
