"""Replay of tests/golden/oracle_vectors.npz (made by tests/golden/make_oracle_vectors.py).

CPU half: the oracle still produces the frozen numbers (bit for bit for the NumPy units; to
1e-5 relative L2 for the torch-CPU blocks, whose float32 summation order depends on the host's
thread count).  GPU half: libpe_hip.so hits the same numbers through the C ABI - bit-exact for
decode, render, loss gradient and the optimizers, and to the block-level tolerances stated in
test_gpu_train.py for the bf16 conv blocks.
"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import oracle
from oracle import torch_oracle as to

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import make_oracle_vectors as mk  # noqa: E402

GOLD = Path(__file__).resolve().parent / "golden" / "oracle_vectors.npz"


@pytest.fixture(scope="module")
def v():
    with np.load(GOLD, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


# ------------------------------------------------------------------ CPU: oracle vs frozen vectors
def test_oracle_decode_matches_golden(v):
    xy, conf, idx = oracle.decode_heatmaps(v["decode_hm"])
    assert np.array_equal(idx, v["decode_idx"])
    assert np.array_equal(xy, v["decode_xy"]) and np.array_equal(conf, v["decode_conf"])
    # the adversarial maps have the answers their construction dictates
    H, W = 64, 48
    assert v["decode_idx"][3, 0] == 0 and np.all(v["decode_xy"][3, 0] == 0)       # all zero
    assert np.all(v["decode_xy"][3, 1] == 0) and v["decode_conf"][3, 1] == -1.0   # all negative
    assert v["decode_idx"][3, 2] == 10 * W + 7                                    # tie
    assert v["decode_idx"][3, 4] == H * W - 1
    assert tuple(v["decode_xy"][3, 5]) == (1.0, 20.0)                             # x == 1
    assert tuple(v["decode_xy"][3, 6]) == (20.25, 30.25)   # plateau: first corner, pulled inwards
    assert tuple(v["decode_xy"][3, 7]) == (25.25, 33.25)


def test_oracle_render_loss_optim_match_golden(v):
    t, w = oracle.render_targets(v["render_joints"], 64, 48, 4.0, 2.0, 6)
    assert np.array_equal(t, v["render_target"]) and np.array_equal(w, v["render_weight"])
    assert w[0, 0] == 0.0 and not t[0, 0].any()
    assert w[0, 2] == 0.0 and w[0, 1] == 1.0 and t[0, 1, 63, 0] == 1.0
    vec, sup = oracle.render_vecmaps(v["render_joints"][:2], mk.VEC_PAIRS, 64, 48, 4.0, 2)
    assert np.array_equal(vec, v["vec_label"])
    assert vec[0, 16:20].any()   # both centres of pair (0, 2) are off the map, windows are not
    vl, vpc, vg = oracle.vecmap_loss_fwd_bwd(v["vec_pred"], v["render_joints"][:2], mk.VEC_PAIRS,
                                             4.0, 2, v["mse_mask"])
    assert np.float32(vl) == v["vec_loss"] and np.array_equal(vpc, v["vec_per_channel"])
    assert np.array_equal(vg, v["vec_grad"])
    loss, pj, grad = oracle.mse_fwd_bwd(v["mse_pred"], t[:2], w[:2], v["mse_mask"])
    assert np.float32(loss) == v["mse_loss"]
    assert np.array_equal(pj, v["mse_per_joint"]) and np.array_equal(grad, v["mse_grad"])
    p, m, s = v["adam_p0"], np.zeros(1000, np.float32), np.zeros(1000, np.float32)
    for step in range(3):
        p, m, s, pb = oracle.adam_step(p, m, s, v["adam_g"][step], 1e-3, 0.9, 0.999, 1e-8, step + 1)
    assert np.array_equal(p, v["adam_p3"]) and np.array_equal(m, v["adam_m3"])
    assert np.array_equal(s, v["adam_v3"]) and np.array_equal(pb, v["adam_pb3"])
    sp, smom, _ = oracle.sgd_step(v["adam_p0"], np.zeros(1000, np.float32), v["adam_g"][0], 0.01, 0.9)
    assert np.array_equal(sp, v["sgd_p1"]) and np.array_equal(smom, v["sgd_mom1"])


def test_oracle_blocks_match_golden(v):
    from pose_estimation_amd import model as pm

    specs, state, bx, bdout, dx_in, dda = mk.block_case()
    blk = specs[1][1]
    out, dx, rg = to.bottleneck_train_reference(blk, state, bx, bdout)
    assert rel_l2(out, v["block_out"]) < 1e-5 and rel_l2(dx, v["block_dx"]) < 1e-5
    for c in (blk.c1, blk.c2, blk.c3):
        packed = pm.pack_weight(c, rg[c.name + ".weight"]).reshape(-1)
        assert rel_l2(packed[::mk.SAMPLE_STEP], v[f"block_{c.name}_dw_sample"]) < 1e-4
    d = specs[2][1]
    a, ddx, dg = to.deconv_train_reference(d, state, dx_in, dda)
    assert rel_l2(a, v["deconv_out"]) < 1e-5 and rel_l2(ddx, v["deconv_dx"]) < 1e-5


# ------------------------------------------------------------------ GPU: HIP path vs frozen vectors
@pytest.mark.gpu
def test_hip_units_match_golden(v):
    import pose_estimation_amd as pe
    from pose_estimation_amd import ops

    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    xy, conf, idx = pe.decode(cu(v["decode_hm"]))
    assert np.array_equal(idx.cpu().numpy(), v["decode_idx"])
    assert np.array_equal(xy.cpu().numpy(), v["decode_xy"])
    assert np.array_equal(conf.cpu().numpy(), v["decode_conf"])
    t, w = ops.render_targets(cu(v["render_joints"]), 64, 48, 4.0, 2.0, 6)
    assert np.array_equal(t.cpu().numpy(), v["render_target"])
    assert np.array_equal(w.cpu().numpy(), v["render_weight"])
    for fused in (False, True):
        kw = (dict(joints=cu(v["render_joints"][:2]), stride=4.0, sigma=2.0, radius=6) if fused
              else dict(target=cu(v["render_target"][:2]), weight=cu(v["render_weight"][:2])))
        loss, pj, grad, _ = ops.mse_fwd_bwd(cu(v["mse_pred"]), mask=cu(v["mse_mask"]), **kw)
        assert loss.item() == pytest.approx(float(v["mse_loss"]), rel=2e-6)
        assert np.allclose(pj.cpu().numpy(), v["mse_per_joint"], rtol=2e-6, atol=0)
        assert np.array_equal(grad.cpu().numpy(), v["mse_grad"])
    jv = cu(v["render_joints"][:2])
    assert np.array_equal(ops.render_vecmaps(jv, mk.VEC_PAIRS, 64, 48, 4.0, 2).cpu().numpy(),
                          v["vec_label"])
    nhwc = ops.render_vecmaps(jv, mk.VEC_PAIRS, 64, 48, 4.0, 2, layout="NHWC")
    assert np.array_equal(nhwc.permute(0, 3, 1, 2).cpu().numpy(), v["vec_label"])
    vl, vpc, vg = ops.vecmap_loss_fwd_bwd(cu(v["vec_pred"]), jv, mk.VEC_PAIRS, 4.0, 2,
                                          mask=cu(v["mse_mask"]))
    assert vl.item() == pytest.approx(float(v["vec_loss"]), rel=2e-6)
    assert np.allclose(vpc.cpu().numpy(), v["vec_per_channel"], rtol=2e-6, atol=0)
    assert np.array_equal(vg.cpu().numpy(), v["vec_grad"])
    p, m, s = cu(v["adam_p0"]), torch.zeros(1000).cuda(), torch.zeros(1000).cuda()
    pb = torch.empty(1000, dtype=torch.bfloat16).cuda()
    for step in range(3):
        ops.adam_step(p, m, s, cu(v["adam_g"][step]), step + 1, lr=1e-3, p_bf16=pb)
    assert np.array_equal(p.cpu().numpy(), v["adam_p3"])
    assert np.array_equal(m.cpu().numpy(), v["adam_m3"])
    assert np.array_equal(s.cpu().numpy(), v["adam_v3"])
    assert np.array_equal(pb.view(torch.int16).cpu().numpy().view(np.uint16), v["adam_pb3"])
    p, mom = cu(v["adam_p0"]), torch.zeros(1000).cuda()
    ops.sgd_step(p, mom, cu(v["adam_g"][0]), 0.01, 0.9)
    assert np.array_equal(p.cpu().numpy(), v["sgd_p1"])
    assert np.array_equal(mom.cpu().numpy(), v["sgd_mom1"])


@pytest.mark.gpu
def test_hip_blocks_match_golden(v):
    from pose_estimation_amd.training import TrainablePoseResNet

    specs, state, bx, bdout, dx_in, dda = mk.block_case()
    net = TrainablePoseResNet("cuda:0", layers=(2, 1, 1, 1))
    net.load_state(state)
    blk = net.specs[1][1]
    net.grads.zero_()
    s = net.block_forward(blk, bx.to(torch.bfloat16).cuda())
    dx = net.block_backward(blk, s, bdout.to(torch.bfloat16).cuda())
    out = s["out"].float().cpu().numpy()
    assert np.abs(out - v["block_out"]).max() <= 8e-3 * np.abs(v["block_out"]).max()  # one bf16 ulp
    assert rel_l2(dx.float().cpu(), v["block_dx"]) <= 2e-2
    for c in (blk.c1, blk.c2, blk.c3):
        got = net.view(net.grads, c.name + ".weight").cpu().reshape(-1)[::mk.SAMPLE_STEP]
        assert rel_l2(got, v[f"block_{c.name}_dw_sample"]) <= 2e-2
        assert rel_l2(net.view(net.grads, c.name + ".bn.gamma").cpu(), v[f"block_{c.name}_dgamma"]) <= 2e-2
        assert rel_l2(net.view(net.grads, c.name + ".bn.beta").cpu(), v[f"block_{c.name}_dbeta"]) <= 2e-2
    d = net.specs[2][1]
    net.grads.zero_()
    sd = net.deconv_forward(d, dx_in.to(torch.bfloat16).cuda())
    ddx = net.deconv_backward(d, sd, dda.to(torch.bfloat16).cuda())
    a = sd["a"].float().cpu().numpy()
    assert np.abs(a - v["deconv_out"]).max() <= 8e-3 * np.abs(v["deconv_out"]).max()
    assert rel_l2(ddx.float().cpu(), v["deconv_dx"]) <= 2e-2
    got = net.view(net.grads, d.name + ".weight").cpu().reshape(-1)[::mk.SAMPLE_STEP]
    assert rel_l2(got, v["deconv_dw_sample"]) <= 2e-2
    assert rel_l2(net.view(net.grads, d.name + ".bn.gamma").cpu(), v["deconv_dgamma"]) <= 2e-2
    assert rel_l2(net.view(net.grads, d.name + ".bn.beta").cpu(), v["deconv_dbeta"]) <= 2e-2


# ============================================================ tape-graph networks (graph_vectors.npz)
import make_graph_vectors as mg  # noqa: E402

GRAPH_GOLD = Path(__file__).resolve().parent / "golden" / "graph_vectors.npz"


@pytest.fixture(scope="module")
def gv():
    with np.load(GRAPH_GOLD, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def test_oracle_graph_blocks_match_golden(gv):
    """One HRNet-style exchange module and two MobileNetV2 inverted-residual blocks: the torch-CPU
    oracle still gives the frozen outputs, input gradients and parameter gradients."""
    spec, state, m, xs, douts = mg.hrnet_case()
    outs, dxs, rg = to.hrnet_module_train_reference(m, state, xs, douts)
    assert len(outs) == 3 and len(rg) > 0
    for i, (o, d) in enumerate(zip(outs, dxs)):
        assert rel_l2(o, gv[f"hr_out{i}"]) < 1e-5 and rel_l2(d, gv[f"hr_dx{i}"]) < 1e-4
    for nm in rg:
        flat = rg[nm].reshape(-1)
        got = flat[::mg.SAMPLE_STEP] if nm.endswith(".weight") else flat
        assert rel_l2(got, gv["hr_g_" + nm]) < 1e-4, nm
        assert flat.norm().item() == pytest.approx(float(gv["hr_n_" + nm]), rel=1e-4)
    blocks, mstate, cases = mg.mobilenet_case()
    assert [b.residual for b, _, _ in cases] == [True, False]
    assert [b.stride for b, _, _ in cases] == [1, 2]
    for b, x, dout in cases:
        out, dx, g = to.mobilenet_block_train_reference(b, mstate, x, dout)
        assert rel_l2(out, gv[f"mb_{b.name}_out"]) < 1e-5
        assert rel_l2(dx, gv[f"mb_{b.name}_dx"]) < 1e-4
        for nm in g:
            assert rel_l2(g[nm].reshape(-1), gv["mb_g_" + nm]) < 1e-4, nm


@pytest.mark.gpu
def test_hip_hrnet_module_matches_golden(gv):
    """The tape graph's exchange module through libpe_hip.so against the frozen oracle vectors:
    outputs within one bf16 ulp of the largest value, input gradients 2e-2 relative L2 (the bounds
    of test_gpu_hrnet_train.py); parameter gradients 3e-2: the widest branch has 96 cells per
    channel here, where a single activation that rounds to the other side of zero moves that
    channel's batch-norm sums by a percent."""
    from pose_estimation_amd import hrnet
    from pose_estimation_amd.graph import Var

    spec, state, m, xs, douts = mg.hrnet_case()
    net = hrnet.TrainablePoseHRNet("cuda:0", num_joints=17, **mg.HR_CFG)
    net.load_state(state)
    m = net.spec.stages[mg.HR_STAGE][0]
    net.grads.zero_()
    net.begin(True)
    xv = [Var(x.to(torch.bfloat16).cuda(), True) for x in xs]
    outs = net._module(m, xv)
    for o, d in zip(outs, douts):
        o.grad = d.to(torch.bfloat16).cuda()
    net.backward({})
    torch.cuda.synchronize()
    for i, (o, x) in enumerate(zip(outs, xv)):
        ref = gv[f"hr_out{i}"]
        assert np.abs(o.t.float().cpu().numpy() - ref).max() <= 8e-3 * np.abs(ref).max() + 1e-6
        assert rel_l2(x.grad.float().cpu(), gv[f"hr_dx{i}"]) <= 2e-2
    checked = 0
    for key in gv:
        if not key.startswith("hr_g_"):
            continue
        nm = key[len("hr_g_"):]
        got = net.view(net.grads, nm).cpu()
        assert got.norm().item() == pytest.approx(float(gv["hr_n_" + nm]), rel=2e-2), nm
        if not nm.endswith(".weight"):
            assert rel_l2(got.reshape(-1), gv[key]) <= 3e-2, nm
        checked += 1
    assert checked >= 20


@pytest.mark.gpu
def test_hip_mobilenet_blocks_match_golden(gv):
    """Inverted-residual blocks (stride 1 with the residual, stride 2) through libpe_hip.so against
    the frozen oracle vectors; bounds of test_gpu_mobilenet_train.py."""
    from pose_estimation_amd import mobilenet as mb
    from pose_estimation_amd.graph import Var

    blocks, mstate, cases = mg.mobilenet_case()
    net = mb.TrainablePoseMobileNetV2("cuda:0", num_joints=17, pairs=None, setting=mg.MB_SETTING)
    net.load_state(mstate)
    for (b0, x, dout), b in zip(cases, [net.blocks[i] for i in mg.MB_BLOCKS]):
        assert b.name == b0.name
        xp = torch.zeros(*x.shape[:3], mb.pad8(b.cin))
        xp[..., :b.cin] = x
        dp = torch.zeros(*dout.shape[:3], mb.pad8(b.cout))
        dp[..., :b.cout] = dout
        net.grads.zero_()
        net.begin(True)
        xv = Var(xp.to(torch.bfloat16).cuda(), True)
        out = net.block(b, xv)
        out.grad = dp.to(torch.bfloat16).cuda()
        net.backward({})
        torch.cuda.synchronize()
        ref = gv[f"mb_{b.name}_out"]
        o = out.t.float().cpu().numpy()
        assert not o[..., b.cout:].any()
        assert np.abs(o[..., :b.cout] - ref).max() <= 8e-3 * np.abs(ref).max() + 1e-6
        dx = xv.grad.float().cpu()
        assert not dx[..., b.cin:].any()
        assert rel_l2(dx[..., :b.cin], gv[f"mb_{b.name}_dx"]) <= 2e-2
        for key in gv:
            nm = key[len("mb_g_"):]
            if not key.startswith("mb_g_" + b.name + "."):
                continue
            got = net.view(net.grads, nm).cpu()
            n = gv[key].size
            if nm.endswith(".weight"):
                L = net.by_name[nm.rsplit(".", 1)[0]]
                if L.kind == "dw":                       # [9, Cpad] -> [C, 1, 3, 3]
                    c = n // 9
                    got = got[:, :c].t().reshape(-1)
                else:                                    # [Cout_pad, 1, 1, Cin_pad] -> [Cout, Cin]
                    co = {"expand": b.mid, "project": b.cout}.get(nm.split(".")[1], None)
                    co = co or got.shape[0]
                    got = got[:co, 0, 0, :n // co].reshape(-1)
            else:
                got = got[:n]
            assert rel_l2(got, gv[key]) <= 2e-2, nm
