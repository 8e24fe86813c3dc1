#!/usr/bin/env python3
"""Times single conv shapes through the C ABI (forward implicit GEMM), HIP events around a loop.

    python tools/conv_bench.py B,H,W,Cin,Cout,k[,stride] [more shapes ...] [--iters N]

Prints one line per shape: microseconds per launch, TFLOP/s and the HBM bytes/s the layer would
need if every tensor crossed HBM once.
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from pose_estimation_amd import nn as pnn  # noqa: E402


import os
RELU = int(os.environ.get("PE_BENCH_RELU", "1"))


def bench(shape, iters):
    B, H, W, Cin, Cout, k = shape[:6]
    stride = shape[6] if len(shape) > 6 else 1
    pad = k // 2
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    # several input buffers so a small layer is not served from cache by construction
    nbuf = 3
    xs = [torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(dev) for _ in range(nbuf)]
    wf = torch.randn((Cout, Cin, k, k), generator=g) * 0.05
    narrow = Cin in (8, 16, 32)
    w = (pnn.pack_conv_weight_narrow(wf) if narrow else pnn.pack_conv_weight(wf)).to(dev)
    kw = {"kw": k} if narrow else {}
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=dev)
    for i in range(3):
        pnn.conv2d(xs[i % nbuf], w, stride=stride, pad=pad, relu=RELU, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        pnn.conv2d(xs[i % nbuf], w, stride=stride, pad=pad, relu=RELU, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flops = 2.0 * B * Ho * Wo * Cout * Cin * k * k
    byts = 2.0 * (xs[0].numel() + out.numel() + w.numel())
    print(f"{','.join(map(str, shape)):28s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  "
          f"{byts / us / 1e6:6.2f} TB/s min-traffic", flush=True)


def main():
    args = sys.argv[1:]
    iters = 20
    if "--iters" in args:
        i = args.index("--iters")
        iters = int(args[i + 1])
        del args[i:i + 2]
    for a in args:
        bench(tuple(int(v) for v in a.split(",")), iters)


if __name__ == "__main__":
    main()
