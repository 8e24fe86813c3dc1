"""Value checks of the kernel variants that only the benchmark's sizes reach.

* ResNet-50 inference at 256x192, batch 256 (BASELINE configs[2], exactly what bench.py times):
  outputs of 64 MB and more leave through the non-temporal epilogue, their residuals arrive
  through non-temporal loads, the 64-channel layers run the 256-pixel x 64-channel workgroup and
  the skip-path projection runs on its side stream.  Sixteen images spread over the batch are
  held against the torch-CPU restatement (same spread-based bound as test_gpu_model.py) and
  must equal, bit for bit, the same images run at batch 1 (other tile shapes, no non-temporal
  path): per output element the K order of the accumulation does not depend on the tile.
* The non-temporal paths forced onto small tensors in a child process (the thresholds are read
  once per process), integer operands, bit-exact: tests/nt_path_child.py.
* The HRNet-W32 network at full width: one exchange module of the last stage at the 384x288
  configuration's sizes (96x72 / 48x36 / 24x18 / 12x9 branch maps), batch 8, output and input
  gradients against torch-CPU autograd of the same module.
* pe_decode_f32 on maps that hold NaN: the documented rule (a NaN never wins, a NaN neighbour
  does not move the coordinate), against the oracle's statement of that rule.
The reference holds none of this arithmetic: parity unpinned.
"""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import oracle
import pose_estimation_amd as pe
from oracle import torch_oracle as to
from pose_estimation_amd import hrnet
from pose_estimation_amd import model as pm
from pose_estimation_amd import nn as pnn
from pose_estimation_amd import synthetic

pytestmark = pytest.mark.gpu
TOL = 4e-3


def test_resnet50_batch_256_predict_against_the_oracle_and_batch_1():
    B, H, W = 256, 256, 192
    net = pm.PoseResNet("cuda:0", num_joints=17, layers=(3, 4, 6, 3))
    state = pm.init_state(net.specs, seed=21)
    pm.randomize_bn(state, seed=22)
    for k in state:      # head weights at a scale that keeps the heat maps O(1)
        if k.startswith("deconv") and k.endswith(".weight"):
            state[k] = state[k] * 30.0
        if k == "final.weight":
            state[k] = state[k] * 100.0
    net.load_state(state)
    img = synthetic.images_u8(B, H, W, seed=5)
    dimg = torch.from_numpy(img).cuda()
    got = net.predict(dimg)
    assert tuple(got.shape) == (B, 17, H // 4, W // 4)
    again = net.predict(dimg)
    assert torch.equal(got, again)
    pick = sorted({0, 1, 17, 31, 63, 64, 100, 127, 128, 129, 170, 191, 200, 222, 254, 255})
    assert len(pick) == 16
    # bit identity with batch 1 (64 x 64 tiles, ring pipelines, no non-temporal path)
    for b in pick[::3]:
        one = net.predict(dimg[b:b + 1].contiguous())
        assert torch.equal(one[0], got[b]), f"image {b} differs between batch 256 and batch 1"
    sub = img[pick]
    ref32 = to.pose_resnet_forward(net.specs, state, sub, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
    ref = to.pose_resnet_forward(net.specs, state, sub, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD,
                                 acc64=True)
    g = got[pick].cpu()
    scale = ref.abs().max().item()
    spread = (ref32 - ref).abs().max().item()
    spread_l2 = ((ref32 - ref).norm() / ref.norm()).item()
    err = (g - ref).abs().max().item()
    err_l2 = ((g - ref).norm() / ref.norm()).item()
    print(f"max|ref| {scale:.3e}  spread {spread:.3e} ({spread_l2:.2e} L2)  "
          f"gpu err {err:.3e} ({err_l2:.2e} L2)")
    assert scale > 1e-3, "degenerate reference output"
    bound = max(3 * spread, TOL * scale)
    assert err <= bound, f"max err {err:.3e} vs bound {bound:.3e}"
    assert err_l2 <= max(3 * spread_l2, TOL)
    # arg-max agreement wherever the reference's best two values are further apart than the bound
    _, _, idx = pe.decode(got[pick].contiguous())
    _, _, oidx = oracle.decode_heatmaps(ref.numpy())
    flat = ref.reshape(ref.shape[0], ref.shape[1], -1).numpy()
    top2 = np.sort(flat, axis=2)[:, :, -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 2 * bound
    same = idx.cpu().numpy() == oidx
    print(f"arg-max: {int(same.sum())} of {same.size} maps agree, {int(clear.sum())} have a clear "
          "maximum in the reference")
    assert same[clear].all()
    # and the decode of the GPU heat maps is bit-exact against the oracle decode of the same maps
    gxy, gconf, gidx = oracle.decode_heatmaps(g.numpy())
    xy, conf, _ = pe.decode(got[pick].contiguous())
    assert np.array_equal(idx.cpu().numpy(), gidx) and np.array_equal(xy.cpu().numpy(), gxy)
    assert np.array_equal(conf.cpu().numpy(), gconf)


def test_non_temporal_paths_forced_on_small_tensors():
    child = Path(__file__).resolve().parent / "nt_path_child.py"
    env = dict(os.environ, PE_IGEMM_NT_MB="1", PE_DW_NT_MB="1")
    r = subprocess.run([sys.executable, str(child)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "NT_PATH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def rel_l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


# Bounds of the full-width module test.  Between a branch input and a module output lie four
# residual units (eight convs, eight batch norms, four bf16-rounded residual sums) and the
# fusion.  An accumulation-order difference flips an occasional bf16 rounding (and with it an
# occasional ReLU mask in backward) and the units behind it carry the flip along, so the CPU
# restatement is run twice, with float32 and with float64 accumulation in its convolutions:
# the distance between the two is what summation order alone does to this module, and the
# MFMA order is a third order.  Each quantity is held to max(floor, 3 x that spread), the rule
# the end-to-end forward tests use; the floors are the reduced-depth module test's bounds.
OUT_MAX, OUT_L2, GRAD_L2 = 8e-3, 4e-3, 2e-2


def test_full_width_hrnet_exchange_module_at_384x288_sizes():
    """Last stage (four branches, widths 32/64/128/256, four residual units each) of HRNet-W32
    at the branch sizes of a 384x288 input, batch 8, against torch-CPU autograd."""
    from pose_estimation_amd.graph import Var
    from pose_estimation_amd.model import randomize_bn

    net = hrnet.TrainablePoseHRNet("cuda:0", num_joints=17)
    state = hrnet.init_state(net.spec, seed=31)
    randomize_bn(state, seed=32)
    net.load_state(state)
    g = torch.Generator().manual_seed(33)
    B, H, W = 8, 96, 72
    m = net.spec.stages[-1][0]
    nb = len(m.branches)
    assert nb == 4 and net.spec.widths[0] == 32
    widths = net.spec.widths[:nb]
    xs = [to.bf16r(torch.randn((B, H >> i, W >> i, widths[i]), generator=g)) for i in range(nb)]
    net.grads.zero_()
    net.begin(True)
    xv = [Var(x.to(torch.bfloat16).cuda(), True) for x in xs]
    outs = net._module(m, xv)
    douts = [to.bf16r(torch.randn(tuple(o.t.shape), generator=g)) for o in outs]
    for o, d in zip(outs, douts):
        o.grad = d.to(torch.bfloat16).cuda()
    net.backward({})
    torch.cuda.synchronize()
    routs, rdxs, rg = to.hrnet_module_train_reference(m, state, xs, douts, acc64=True)
    souts, sdxs, sg = to.hrnet_module_train_reference(m, state, xs, douts)

    def packed(nm, t, like):
        L = net.by_name[nm.rsplit(".", 1)[0]] if nm.endswith(".weight") else None
        return (L.pack(t) if L is not None else t).reshape(like.shape)

    rows = []      # (what, gpu error, the restatements' own spread, floor)
    for i, (o, r, s_) in enumerate(zip(outs, routs, souts)):
        got = o.t.float().cpu()
        rows.append((f"out[{i}] max", (got - r).abs().max().item() / r.abs().max().item(),
                     (s_ - r).abs().max().item() / r.abs().max().item(), OUT_MAX))
        rows.append((f"out[{i}] L2", rel_l2(got, r), rel_l2(s_, r), OUT_L2))
    for i, (v, r, s_) in enumerate(zip(xv, rdxs, sdxs)):
        rows.append((f"dx[{i}] L2", rel_l2(v.grad.float().cpu(), r), rel_l2(s_, r), GRAD_L2))
    assert len(rg) > 50
    for nm, refg in rg.items():
        got = net.view(net.grads, nm).cpu()
        rows.append((nm, rel_l2(got, packed(nm, refg, got)),
                     rel_l2(packed(nm, sg[nm], got), packed(nm, refg, got)), GRAD_L2))
    bad = []
    for what, err, spread, floor in rows:
        flag = "" if err <= max(floor, 3 * spread) else "   <-- over the bound"
        if flag or not what.startswith("stage"):
            print(f"{what:40s} gpu {err:.2e}   f32-vs-f64 restatement {spread:.2e}{flag}")
        if flag:
            bad.append(what)
    pg = [r for r in rows if r[0].startswith("stage")]
    w = max(pg, key=lambda r: r[1])
    print(f"parameter gradients: worst gpu {w[1]:.2e} ({w[0]}, spread {w[2]:.2e}) over {len(pg)}")
    assert not bad, bad


def test_decode_nan_rule():
    rng = np.random.default_rng(3)
    B, K, H, W = 2, 6, 16, 12
    hm = rng.uniform(0.0, 0.5, (B, K, H, W)).astype(np.float32)
    hm[0, 0, 5, 6] = 1.0
    hm[0, 0, 0, 0] = np.nan                       # a NaN before the maximum
    hm[0, 1, 5, 6] = 1.0
    hm[0, 1, 9, 9] = np.nan                       # a NaN after it
    hm[0, 2, 5, 6] = 1.0
    hm[0, 2, 5, 7] = np.nan                       # NaN neighbour: x must not move
    hm[0, 3, 5, 6] = 1.0
    hm[0, 3, 4, 6] = np.nan                       # NaN neighbour above: y must not move
    hm[0, 4] = np.nan                             # nothing but NaN
    hm[0, 5] = -np.inf
    hm[0, 5, 3, 3] = np.nan                       # NaN and -inf only
    hm[1, 0, :, ::2] = np.nan                     # every other column
    hm[1, 1, 15, 11] = np.nan                     # the very last element
    hm[1, 2] = np.where(rng.uniform(size=(H, W)) < 0.3, np.nan, hm[1, 2]).astype(np.float32)
    want = oracle.decode_heatmaps_nan_skipped(hm)
    for layout in ("NCHW", "NHWC", "strided"):
        if layout == "NCHW":
            got = pe.decode(torch.from_numpy(hm).cuda())
        elif layout == "NHWC":
            got = pe.decode(torch.from_numpy(np.ascontiguousarray(hm.transpose(0, 2, 3, 1))).cuda(),
                            layout="NHWC")
        else:
            pad = torch.zeros(B, K + 1, H, W + 2, device="cuda")
            pad[:, :K, :, :W] = torch.from_numpy(hm).cuda()
            got = pe.decode(pad[:, :K, :, :W])
        xy, conf, idx = (t.cpu().numpy() for t in got)
        assert np.array_equal(idx, want[2]), layout
        assert np.array_equal(xy, want[0]), layout
        assert np.array_equal(np.isnan(conf), np.isnan(want[1])), layout
        ok = ~np.isnan(conf)
        assert np.array_equal(conf[ok], want[1][ok]), layout
