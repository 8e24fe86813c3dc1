#!/usr/bin/env python3
"""Achieved bandwidth of the batch-norm kernels at the activation sizes of a batch-64 training
step.      python tools/bn_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pose_estimation_amd import nn as pnn  # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    dev = "cuda"
    for (B, H, W, C) in [(64, 64, 48, 256), (64, 64, 48, 64), (64, 32, 24, 512), (64, 32, 24, 128),
                         (64, 16, 12, 1024), (64, 16, 12, 256), (64, 8, 6, 2048), (64, 8, 6, 512)]:
        M = B * H * W
        # rotate over enough buffers that the 256 MB last-level cache cannot serve them
        nbuf = max(2, int(600e6 // (M * C * 2 * 4)) + 1)
        zs = [torch.randn(B, H, W, C, device=dev).to(BF16) for _ in range(nbuf)]
        das = [torch.randn(B, H, W, C, device=dev).to(BF16) for _ in range(nbuf)]
        as_ = [torch.relu(torch.randn(B, H, W, C, device=dev)).to(BF16) for _ in range(nbuf)]
        outs = [torch.empty(B, H, W, C, device=dev, dtype=BF16) for _ in range(nbuf)]
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ws = pnn.bn_workspace(M, C, dev)
        st = pnn.bn_stats(zs[0], gamma, beta, 1e-5, 0.1, None, None, ws)
        mean, rstd, scale, shift = st
        it = [0]

        def nxt():
            it[0] = (it[0] + 1) % nbuf
            return it[0]

        nbytes = M * C * 2
        t_stats = timeit(lambda: pnn.bn_stats(zs[nxt()], gamma, beta, 1e-5, 0.1, None, None, ws))
        t_apply = timeit(lambda: pnn.bn_apply(zs[nxt()], scale, shift, None, 1, out=outs[it[0]]))
        t_bwd = timeit(lambda: pnn.bn_bwd(das[nxt()], as_[it[0]], zs[it[0]], gamma, mean, rstd, dg, db,
                                          ws, dz=outs[it[0]]))
        print(f"{B}x{H}x{W}x{C:5d} ({nbytes / 1e6:6.1f} MB): stats {t_stats:6.1f} us "
              f"{nbytes / t_stats / 1e6:5.2f} TB/s | apply {t_apply:6.1f} us {2 * nbytes / t_apply / 1e6:5.2f} TB/s"
              f" | bwd(3 kernels) {t_bwd:6.1f} us {7 * nbytes / t_bwd / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
