"""conv2 (3x3 64 -> 64) + conv3 (1x1 64 -> 256, residual, ReLU) of a first-stage ResNet bottleneck at
batch 256: the fused launch (pe_conv3x3_c64_pw_fwd_bf16) against the two launches it replaces.

    python tools/c2c3_bench.py
"""
import sys, torch
sys.path.insert(0, ".")
from pose_estimation_amd import nn as pnn
BF16 = torch.bfloat16
B, H, W = 256, 64, 48
g = torch.Generator().manual_seed(1)
x = torch.randn((B, H, W, 64), generator=g).relu().to(BF16).cuda()
w = (torch.randn((64, 3, 3, 64), generator=g) * 0.06).to(BF16).cuda()
w3 = (torch.randn((256, 1, 1, 64), generator=g) * 0.17).to(BF16).cuda()
s = torch.ones(64).cuda(); h = torch.zeros(64).cuda()
s3 = torch.ones(256).cuda(); h3 = torch.zeros(256).cuda()
res = [torch.randn((B, H, W, 256), generator=g).to(BF16).cuda() for _ in range(2)]
outs = [torch.empty((B, H, W, 256), dtype=BF16).cuda() for _ in range(2)]
mid = torch.empty((B, H, W, 64), dtype=BF16).cuda()
def fused(i): pnn.conv3x3_c64_pw(x, w, s, h, w3, s3, h3, res[i & 1], True, out=outs[i & 1])
def sep(i):
    pnn.conv2d(x, w, s, h, None, 1, 1, True, out=mid)
    pnn.conv2d(mid, w3, s3, h3, res[i & 1], 1, 0, True, out=outs[i & 1])
wd = (torch.randn((256, 1, 1, 64), generator=g) * 0.17).to(BF16).cuda()
x0 = torch.randn((B, H, W, 64), generator=g).relu().to(BF16).cuda()
def proj(i): pnn.conv3x3_c64_pw_proj(x, w, s, h, w3, s3, h3, x0, wd, s3, h3, out=outs[i & 1])
def sep_proj(i):
    pnn.conv2d(x0, wd, s3, h3, None, 1, 0, False, out=res[i & 1])
    fused(i)
w1 = (torch.randn((64, 1, 1, 256), generator=g) * 0.09).to(BF16).cuda()
nxt = torch.empty((B, H, W, 64), dtype=BF16).cuda()
def fused_next(i):
    pnn.conv3x3_c64_pw_next(x, w, s, h, w3, s3, h3, res[i & 1], w1, s, h, out=outs[i & 1], out_next=nxt)
def fused_then_next(i):
    fused(i)
    pnn.conv2d(outs[i & 1], w1, s, h, None, 1, 0, True, out=nxt)
cases = (("fused+next conv1", fused_next), ("fused+next conv1", fused_next)) if "--next" in sys.argv else None
for name, fn in cases or (("fused+next conv1", fused_next), ("fused, then next conv1", fused_then_next),
                 ("fused+next conv1", fused_next), ("fused, then next conv1", fused_then_next),
                 ("fused", fused), ("separate", sep), ("fused", fused), ("separate", sep),
                 ("fused+projection", proj), ("projection, then fused", sep_proj),
                 ("fused+projection", proj), ("projection, then fused", sep_proj)):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for i in range(30): fn(i)
    e1.record(); torch.cuda.synchronize()
    print(name, "%.1f us" % (e0.elapsed_time(e1) / 30 * 1e3))
