"""Child process of test_gpu_bench_paths.py: the non-temporal epilogue of the implicit GEMM and
the non-temporal store path of the stride-1 depthwise kernel, forced onto small tensors.

The library reads PE_IGEMM_NT_MB / PE_DW_NT_MB once per process, so the thresholds are set in
this process's environment before the library is loaded (both to 1 MB; every output below is
larger).  Integer operands make every product and partial sum exact in float32, so the bf16
results must equal the torch-CPU restatement bit for bit.  Prints "NT_PATH_OK" on success.
"""
import os
import sys
from pathlib import Path

os.environ["PE_IGEMM_NT_MB"] = "1"
os.environ["PE_DW_NT_MB"] = "1"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

from oracle import torch_oracle as to  # noqa: E402
from pose_estimation_amd import nn as pnn  # noqa: E402

BF16 = torch.bfloat16


def ints(shape, g, lo, hi):
    return torch.randint(lo, hi + 1, shape, generator=g).to(torch.float32)


def conv_case(B, H, W, cin, cout, k, stride, seed):
    """conv + per-channel scale/shift (integers) + residual + ReLU; output >= 1 MB."""
    g = torch.Generator().manual_seed(seed)
    p = k // 2
    x = ints((B, H, W, cin), g, -2, 2)
    w = ints((cout, cin, k, k), g, -1, 1)
    scale = ints((cout,), g, 1, 2)
    shift = ints((cout,), g, -3, 3)
    Ho, Wo = (H + 2 * p - k) // stride + 1, (W + 2 * p - k) // stride + 1
    res = ints((B, Ho, Wo, cout), g, -4, 4)
    assert B * Ho * Wo * cout * 2 >= 1_000_000, "output below the forced threshold"
    ref = to.bf16r(to.conv2d(x, w, scale, shift, res, stride, p, relu=True))
    got = pnn.conv2d(x.to(BF16).cuda(), pnn.pack_conv_weight(w).cuda(), scale.cuda(), shift.cuda(),
                     res.to(BF16).cuda(), stride, p, relu=True)
    assert torch.equal(got.float().cpu(), ref), ("conv", B, H, W, cin, cout, k, stride)
    # and without a residual (non-temporal stores only)
    ref2 = to.bf16r(to.conv2d(x, w, stride=stride, pad=p))
    got2 = pnn.conv2d(x.to(BF16).cuda(), pnn.pack_conv_weight(w).cuda(), stride=stride, pad=p)
    assert torch.equal(got2.float().cpu(), ref2), ("conv plain", B, H, W, cin, cout, k, stride)


def deconv_case(B, H, W, cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = ints((B, H, W, cin), g, -2, 2)
    w = ints((cin, cout, 4, 4), g, -1, 1)
    assert B * 4 * H * W * cout * 2 >= 1_000_000
    ref = to.bf16r(to.deconv4x4s2(x, w, relu=True))
    got = pnn.deconv4x4s2(x.to(BF16).cuda(), pnn.pack_deconv_weight(w).cuda(), relu=True)
    assert torch.equal(got.float().cpu(), ref), ("deconv", B, H, W, cin, cout)


def depthwise_case(B, H, W, C, dil, seed):
    g = torch.Generator().manual_seed(seed)
    x = ints((B, H, W, C), g, -2, 2)
    w = ints((C, 1, 3, 3), g, -1, 1)
    assert B * H * W * C * 2 >= 1_000_000
    ref = to.bf16r(to.depthwise3x3(x, w, None, None, 1, dil, relu6=True))
    w9c = w[:, 0].reshape(C, 9).t().contiguous().to(BF16).cuda()
    got = pnn.depthwise3x3(x.to(BF16).cuda(), w9c, None, None, 1, dil, 2)
    assert torch.equal(got.float().cpu(), ref), ("depthwise", B, H, W, C, dil)


def main():
    assert torch.cuda.is_available()
    conv_case(3, 40, 36, 64, 256, 1, 1, seed=1)     # 128 x 128 tiles, residual + ReLU
    conv_case(5, 33, 29, 128, 128, 3, 1, seed=2)    # 3x3, part-filled last tile
    conv_case(40, 64, 48, 64, 64, 3, 1, seed=3)     # 1280 tiles of 128: the 256 x 64 workgroup
    conv_case(4, 48, 36, 256, 512, 1, 2, seed=4)    # strided projection
    conv_case(4, 30, 30, 96, 160, 1, 1, seed=5)     # part-filled K step, N not a tile multiple
    deconv_case(3, 16, 12, 256, 256, seed=6)        # four phases scattered through nt stores
    depthwise_case(2, 40, 40, 192, 1, seed=7)
    depthwise_case(2, 40, 40, 384, 2, seed=8)
    torch.cuda.synchronize()
    print("NT_PATH_OK")


if __name__ == "__main__":
    main()
