"""HRNet-style multi-branch pose path: narrow-input convs (8/16/32 channels as multi-pixel K
runs), the upsample-add fusion kernel and the whole network against the torch-CPU restatement
(same bf16 rounding points, float32 accumulation).  The reference holds no such network:
parity unpinned.

Single layers: |d| <= 4e-3 * max|ref| (one bf16 rounding of the output plus float32 summation
order).  Whole network, two checks: (1) every conv unit of a forward pass, re-computed by the
restatement from the very input the GPU unit saw, to that same single-layer bound; (2) end to
end, max(3 x the float32-vs-float64 accumulation spread of the restatement, 2e-2 * max|ref|):
an accumulation-order difference flips an occasional bf16 rounding by one ulp (0.4-0.8 %), and
the later layers of a random-weight network carry such flips to the output undamped.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import torch_oracle as to
from pose_estimation_amd import hrnet
from pose_estimation_amd import nn as pnn
from pose_estimation_amd import synthetic
from pose_estimation_amd.model import randomize_bn

BF16 = torch.bfloat16
TOL = 4e-3


def rnd(shape, g, s=1.0):
    return to.bf16r(torch.randn(shape, generator=g) * s)


def close(got, ref, tol=TOL):
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert scale > 0 and err <= tol * scale, f"max err {err:.3e} vs {tol * scale:.3e}"


# ------------------------------------------------------------------ host logic (CPU)
def test_hrnet_w32_layer_list():
    sp = hrnet.hrnet_specs()
    st = hrnet.init_state(sp)
    params = sum(v.numel() for k, v in st.items() if k.endswith(".weight"))
    assert params == pytest.approx(28.5e6, rel=1e-2)              # public HRNet-W32: 28.5 M
    assert [len(m) for m in sp.stages] == [1, 4, 3]
    assert [len(m.branches) for m in sp.stages[2]] == [4, 4, 4]
    assert len(sp.stages[2][-1].fuse) == 1 and len(sp.stages[2][0].fuse) == 4
    chain = sp.stages[2][0].fuse[3][0]                             # stride 4 -> stride 32
    assert [(u.cin, u.cout, u.stride, u.relu) for u in chain] == \
        [(32, 32, 2, True), (32, 32, 2, True), (32, 256, 2, False)]
    # ~16 GMAC at 384x288 in the public tables (which leave out a few small layers)
    assert 15.5e9 < hrnet.forward_macs(sp, 384, 288) < 17.5e9
    assert hrnet.forward_macs(sp, 256, 192) * 9 == hrnet.forward_macs(sp, 384, 288) * 4


def test_narrow_weight_packing():
    g = torch.Generator().manual_seed(1)
    w = torch.randn((5, 32, 3, 3), generator=g)
    p = pnn.pack_conv_weight_narrow(w).float()
    assert p.shape == (5, 3, 2, 64)
    wr = w.to(BF16).float()
    assert torch.equal(p[:, 1, 0, :32], wr[:, :, 1, 0]) and torch.equal(p[:, 1, 0, 32:], wr[:, :, 1, 1])
    assert torch.equal(p[:, 2, 1, :32], wr[:, :, 2, 2]) and not p[:, :, 1, 32:].any()
    assert pnn.pack_conv_weight_narrow(torch.randn(4, 16, 1, 1)).shape == (4, 1, 1, 64)
    assert pnn.pack_conv_weight_narrow(torch.randn(4, 8, 3, 3)).shape == (4, 3, 1, 64)


# ------------------------------------------------------------------ single kernels (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    # B, H, W, Cin, Cout, k, stride, residual, relu
    (2, 12, 9, 32, 32, 3, 1, True, True),
    (2, 13, 10, 32, 64, 3, 2, False, True),
    (3, 8, 6, 32, 256, 3, 2, True, False),
    (2, 24, 18, 32, 32, 1, 1, False, False),
    (1, 9, 7, 16, 64, 3, 1, False, True),
    (2, 9, 8, 8, 32, 3, 1, False, True),
    (64, 96, 72, 32, 32, 3, 1, True, True),      # the stride-4 branch at full size
])
def test_narrow_conv_matches_oracle(cfg):
    B, H, W, Cin, Cout, k, stride, use_res, relu = cfg
    g = torch.Generator().manual_seed(3 + Cin + Cout)
    x = rnd((B, H, W, Cin), g)
    w = rnd((Cout, Cin, k, k), g, (2.0 / (k * k * Cin)) ** 0.5)
    sc = torch.rand(Cout, generator=g) + 0.5
    sh = torch.randn(Cout, generator=g) * 0.2
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = rnd((B, Ho, Wo, Cout), g) if use_res else None
    ref = to.conv2d(x, w, sc, sh, res, stride, k // 2, relu)
    got = pnn.conv2d(x.to(BF16).cuda(), pnn.pack_conv_weight_narrow(w).cuda(), sc.cuda(), sh.cuda(),
                     res.to(BF16).cuda() if use_res else None, stride, k // 2, relu, kw=k)
    assert got.shape == (B, Ho, Wo, Cout)
    close(got.float().cpu(), ref)


@pytest.mark.gpu
def test_narrow_head_conv_to_float_maps():
    g = torch.Generator().manual_seed(8)
    x = rnd((2, 24, 18, 32), g)
    w = rnd((17, 32, 1, 1), g, 0.2)
    bias = torch.randn(17, generator=g) * 0.1
    ref = to.conv2d(x, w, None, bias).permute(0, 3, 1, 2)
    got = pnn.conv2d(x.to(BF16).cuda(), pnn.pack_conv_weight_narrow(w).cuda(), None, bias.cuda(),
                     out_f32_nchw=True, kw=1)
    assert got.shape == (2, 17, 24, 18) and got.dtype == torch.float32
    close(got.cpu(), ref, 1e-5)        # float32 output: accumulation order only


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(2, 16, 8, 32, (1, 2, 3), True, True), (1, 24, 12, 64, (1,), True, False),
                                 (3, 8, 8, 256, (2, 1), False, True), (2, 96, 72, 32, (1, 2, 3), True, True)])
def test_fuse_upsample_add_matches_oracle(cfg):
    B, H, W, C, facs, has_base, relu = cfg
    g = torch.Generator().manual_seed(13 + C)
    base = rnd((B, H, W, C), g) if has_base else None
    lows = [rnd((B, H >> f, W >> f, C), g) for f in facs]
    ref = base.clone() if has_base else torch.zeros((B, H, W, C))
    for t, f in zip(lows, facs):
        ref = ref + t.repeat_interleave(1 << f, dim=1).repeat_interleave(1 << f, dim=2)
    ref = to.bf16r(torch.relu(ref) if relu else ref)
    got = pnn.fuse_upsample_add(base.to(BF16).cuda() if has_base else None,
                                [t.to(BF16).cuda() for t in lows], list(facs), relu)
    assert torch.equal(got.float().cpu(), ref)    # same float32 sum order, one rounding


@pytest.mark.gpu
def test_fuse_argument_errors():
    x = torch.zeros((1, 10, 8, 32), dtype=BF16, device="cuda")
    with pytest.raises(ValueError):
        pnn.fuse_upsample_add(x, [torch.zeros((1, 5, 5, 32), dtype=BF16, device="cuda")], [1])
    with pytest.raises(Exception, match="multiple of the upsampling factor"):
        pnn.fuse_upsample_add(x, [torch.zeros((1, 2, 2, 32), dtype=BF16, device="cuda")], [2])


# ------------------------------------------------------------------ whole network (GPU)
def run_net(B, H, W, seed, **kw):
    net = hrnet.PoseHRNet("cuda:0", **kw)
    st = hrnet.init_state(net.spec, seed=seed)
    randomize_bn(st, seed=seed + 1)
    st["final.weight"] = st["final.weight"] * 100.0
    net.load_state(st)
    img = synthetic.images_u8(B, H, W, seed=1)
    got = net.predict(torch.from_numpy(img).cuda())
    ref32 = to.hrnet_pose_forward(net.spec, st, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
    ref64 = to.hrnet_pose_forward(net.spec, st, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD, acc64=True)
    return net, got, ref32, ref64


def check_maps(got, ref32, ref64):
    g = got.cpu()
    scale = ref64.abs().max().item()
    spread = (ref32 - ref64).abs().max().item()
    err = (g - ref64).abs().max().item()
    print(f"max|ref| {scale:.3e} spread {spread:.3e} gpu err {err:.3e}")
    assert scale > 1e-3
    assert err <= max(3 * spread, 2e-2 * scale)


class RecordingHRNet(hrnet.PoseHRNet):
    """Keeps what every conv unit saw and produced."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.tape = []

    def _unit(self, u, x, residual=None, relu=None):
        y = super()._unit(u, x, residual, relu)
        self.tape.append((u, x, residual, u.relu if relu is None else relu, y))
        return y


def check_units_on_their_own_inputs(kw, B, H, W, seed, every=1):
    net = RecordingHRNet("cuda:0", **kw)
    st = hrnet.init_state(net.spec, seed=seed)
    randomize_bn(st, seed=seed + 1)
    net.load_state(st)
    net.predict(torch.from_numpy(synthetic.images_u8(B, H, W, seed=1)).cuda())
    assert len(net.tape) == len(hrnet.all_units(net.spec)) - 1      # all but the s2d stem conv
    worst = 0.0
    for u, x, res, relu, y in net.tape[::every]:
        sc = st[u.name + ".bn.gamma"] / torch.sqrt(st[u.name + ".bn.var"] + 1e-5)
        sh = st[u.name + ".bn.beta"] - st[u.name + ".bn.mean"] * sc
        ref = to.conv2d(x.float().cpu(), to.bf16r(st[u.name + ".weight"]), sc, sh,
                        res.float().cpu() if res is not None else None, u.stride, u.k // 2, relu)
        scale = ref.abs().max().item()
        err = (y.float().cpu() - ref).abs().max().item()
        assert err <= TOL * scale, f"{u.name}: {err:.3e} vs {TOL * scale:.3e}"
        worst = max(worst, err / scale)
    print(f"{len(net.tape[::every])} units, worst relative error {worst:.2e}")


@pytest.mark.gpu
def test_hrnet_every_unit_matches_oracle_on_its_own_input():
    check_units_on_their_own_inputs(dict(modules=(1, 1, 1), blocks=1, layer1_blocks=1), 2, 64, 64, 5)
    check_units_on_their_own_inputs({}, 1, 128, 96, 6, every=3)


@pytest.mark.gpu
def test_hrnet_forward_reduced_depth():
    net, got, ref32, ref64 = run_net(2, 64, 64, seed=5, modules=(1, 1, 1), blocks=1, layer1_blocks=1)
    assert got.shape == (2, 17, 16, 16)
    check_maps(got, ref32, ref64)


@pytest.mark.gpu
def test_hrnet_w32_forward_full_depth_and_decode():
    net, got, ref32, ref64 = run_net(1, 128, 96, seed=6)
    assert got.shape == (1, 17, 32, 24)
    check_maps(got, ref32, ref64)
    xy, conf, idx = net.decode(got)
    oxy, oconf, oidx = oracle.decode_heatmaps(got.cpu().numpy())
    assert np.array_equal(idx.cpu().numpy(), oidx) and np.array_equal(xy.cpu().numpy(), oxy)


@pytest.mark.gpu
def test_hrnet_384x288_shapes_and_determinism():
    net = hrnet.PoseHRNet("cuda:0")
    net.load_state(hrnet.init_state(net.spec, seed=9))
    img = torch.from_numpy(synthetic.images_u8(2, 384, 288, seed=2)).cuda()
    a = net.predict(img)
    assert a.shape == (2, 17, 96, 72)                              # 96x72 maps at 384x288
    assert torch.equal(a, net.predict(img))
    assert torch.equal(a[1:2], net.predict(img[1:2].contiguous()))


@pytest.mark.gpu
def test_branch_streams_give_the_single_stream_result():
    """Branches and fusion rows of an exchange module run on streams of their own; the kernels
    and their operands are the same, so the heat maps are bit-identical, also over repeated
    passes (freed activations are reused across streams) and different batch sizes."""
    net = hrnet.PoseHRNet("cuda:0")
    net.load_state(hrnet.init_state(net.spec, seed=11))
    assert net.branch_streams
    imgs = [torch.from_numpy(synthetic.images_u8(b, 192, 128, seed=3 + b)).cuda() for b in (3, 1, 4)]
    net.branch_streams = False
    want = [net.predict(im).clone() for im in imgs]
    net.branch_streams = True
    for rep in range(6):
        for im, w in zip(imgs, want):
            assert torch.equal(net.predict(im), w)
    assert len(net._side) == 3
