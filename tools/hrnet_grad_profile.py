"""Prints, layer by layer, the relative L2 error of the HRNet-style training graph's gradients
against the CPU autograd restatement (bf16 rounding points) and that restatement's own
float32-vs-bf16 spread.  usage: python tools/hrnet_grad_profile.py [B H4 W4]"""
import sys

import torch

sys.path.insert(0, ".")
from oracle import torch_oracle as to  # noqa: E402
from pose_estimation_amd import hrnet  # noqa: E402
from pose_estimation_amd.graph import Var  # noqa: E402
from pose_estimation_amd.model import randomize_bn  # noqa: E402

B, H4, W4 = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (6, 64, 48)
cfg = dict(width=32, modules=(1, 1, 1), blocks=1, layer1_blocks=1)
net = hrnet.TrainablePoseHRNet("cuda:0", num_joints=17, **cfg)
state = hrnet.init_state(net.spec, seed=11)
randomize_bn(state, seed=12)
state["final.weight"] = state["final.weight"] * 100.0
net.load_state(state)
g = torch.Generator().manual_seed(13)
x = to.bf16r(torch.randn((B, H4, W4, 64), generator=g))
dout = torch.randn((B, 17, H4, W4), generator=g)
sp = net.spec
net.begin(True)
v = Var(x.to(torch.bfloat16).cuda(), True)
for b in sp.layer1:
    idn = net._u(b.down, v) if b.down else v
    v = net._u(b.c3, net._u(b.c2, net._u(b.c1, v)), residual=idn, relu=True)
xs = [v]
for tr, mods in zip(sp.transitions, sp.stages):
    xs = [(xs[i] if t is None else net._u(t, xs[i])) if i < len(xs) else net._u(t, xs[-1])
          for i, t in enumerate(tr)]
    for m in mods:
        xs = net._module(m, xs)
hm = net.head(net.by_name["final"], xs[0])
net.backward({"final": dout.cuda()})
torch.cuda.synchronize()
_, rhm, rg = to.hrnet_train_reference(sp, state, None, None, None, None, None, None, x_nhwc=x,
                                      dout_nchw=dout)
_, fhm, fg = to.hrnet_train_reference(sp, state, None, None, None, None, None, None, x_nhwc=x,
                                      dout_nchw=dout, round_bf16=False)


def rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


print("heat maps", rel(hm.cpu(), rhm), rel(fhm, rhm))
for L in net.layers:
    if L.name.startswith("stem"):
        continue
    nm = L.name + ".weight"
    got = net.view(net.grads, nm).cpu()
    ref = L.pack(rg[nm]).reshape(got.shape)
    f32 = L.pack(fg[nm]).reshape(got.shape)
    print(f"{nm:36s} gpu-vs-bf16ref {rel(got, ref):.3e}   f32ref-vs-bf16ref {rel(f32, ref):.3e}   "
          f"gpu-vs-f32ref {rel(got, f32):.3e}")
