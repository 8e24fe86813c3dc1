#!/usr/bin/env python3
"""Times the backward-weight GEMM of single conv shapes through the C ABI, HIP events around a
loop.      python tools/wgrad_bench.py [B,H,W,Cin,Cout,k[,stride] ...] [--iters N] [--batched]
Prints microseconds per call (kernel + slab fold) and TFLOP/s."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pose_estimation_amd import nn as pnn  # noqa: E402

DEFAULT = ["64,64,48,64,64,3", "64,32,24,128,128,3", "64,16,12,256,256,3", "64,8,6,512,512,3",
           "64,64,48,256,64,1", "64,64,48,64,256,1", "64,32,24,512,128,1", "64,32,24,128,512,1",
           "64,16,12,1024,256,1", "64,16,12,256,1024,1", "64,8,6,2048,512,1", "64,8,6,512,2048,1",
           "64,64,48,128,128,3,2", "64,32,24,256,256,3,2", "64,64,48,256,512,1,2",
           "64,96,72,32,32,3", "64,48,36,64,64,3", "64,24,18,128,128,3", "64,12,9,256,256,3"]


def bench(shape, iters):
    B, H, W, Cin, Cout, k = shape[:6]
    stride = shape[6] if len(shape) > 6 else 1
    pad = k // 2
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xs = [torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(dev) for _ in range(2)]
    dys = [torch.randn((B, Ho, Wo, Cout), generator=g).to(torch.bfloat16).to(dev) for _ in range(2)]
    narrow = Cin in (8, 16, 32)
    kwc = (k + 64 // Cin - 1) // (64 // Cin) * 64 if narrow else k * Cin
    dw = torch.zeros(Cout * k * kwc, dtype=torch.float32, device=dev)
    for i in range(3):
        pnn.conv2d_bwd_weight(xs[i % 2], dys[i % 2], dw, k, stride, pad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        pnn.conv2d_bwd_weight(xs[i % 2], dys[i % 2], dw, k, stride, pad)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flops = 2.0 * B * Ho * Wo * Cout * Cin * k * k
    print(f"{','.join(map(str, shape)):26s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
    return us, flops


def bench_batched(shapes, iters):
    """All shapes as jobs of batched launches (eight per launch), timed together."""
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    jobs, flops = [], 0.0
    for shape in shapes:
        B, H, W, Cin, Cout, k = shape[:6]
        stride = shape[6] if len(shape) > 6 else 1
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        x = torch.randn((B, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
        dy = torch.randn((B, Ho, Wo, Cout), generator=g).to(torch.bfloat16).to(dev)
        narrow = Cin in (8, 16, 32)
        kwc = (k + 64 // Cin - 1) // (64 // Cin) * 64 if narrow else k * Cin
        jobs.append((x, dy, torch.zeros(Cout * k * kwc, dtype=torch.float32, device=dev), k, stride,
                     pad, None))
        flops += 2.0 * B * Ho * Wo * Cout * Cin * k * k
    for _ in range(3):
        pnn.conv2d_bwd_weight_batched(jobs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        pnn.conv2d_bwd_weight_batched(jobs)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"batched: {len(jobs)} layers in {us:.1f} us  {flops / us / 1e6:.1f} TFLOP/s", flush=True)


def main():
    args = sys.argv[1:]
    iters = 30
    batched = "--batched" in args
    if batched:
        args.remove("--batched")
    if "--iters" in args:
        i = args.index("--iters")
        iters = int(args[i + 1])
        del args[i:i + 2]
    tot_us = tot_fl = 0.0
    for a in (args or DEFAULT):
        us, fl = bench(tuple(int(v) for v in a.split(",")), iters)
        tot_us += us
        tot_fl += fl
    print(f"aggregate {tot_fl / tot_us / 1e6:.1f} TFLOP/s over {tot_us:.0f} us")
    if batched:
        bench_batched([tuple(int(v) for v in a.split(",")) for a in (args or DEFAULT)], iters)


if __name__ == "__main__":
    main()
