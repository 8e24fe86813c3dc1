"""Training of the HRNet-style multi-branch network on the tape graph: forward with batch
statistics, loss, backward through branches, exchange modules and narrow (32-channel) layers,
against torch-CPU autograd of the same graph with bf16 rounding at the kernels' storage points.
The reference holds no such network: parity unpinned.

Bounds, as for the ResNet training test: a deep bf16 network amplifies summation order through
batch-norm backward, so gradients are compared in relative L2 per parameter tensor against
max(3e-2, 3 x the float32-vs-bf16 spread of the CPU graph itself); heat maps and loss likewise.
The kernels under the graph are pinned bit for bit in test_gpu_bwd.py.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import torch_oracle as to
from pose_estimation_amd import hrnet
from pose_estimation_amd import nn as pnn
from pose_estimation_amd import ops
from pose_estimation_amd import synthetic
from pose_estimation_amd.model import randomize_bn

pytestmark = pytest.mark.gpu
SMALL = dict(width=32, modules=(1, 1, 1), blocks=1, layer1_blocks=1)


def rel_l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def make_case(B, H, W, seed, cfg=SMALL, sigma=2.0):
    net = hrnet.TrainablePoseHRNet("cuda:0", num_joints=17, sigma=sigma, **cfg)
    state = hrnet.init_state(net.spec, seed=seed)
    randomize_bn(state, seed=seed + 1)
    state["final.weight"] = state["final.weight"] * 100.0   # O(1) heat maps
    net.load_state(state)
    img = synthetic.images_u8(B, H, W, seed=1)
    joints = synthetic.joints(B, 17, H, W, seed=2)
    mask = synthetic.ignore_mask(B, H // 4, W // 4, seed=4)
    return net, state, img, joints, mask


def packed(net, name, ref):
    L = net.by_name[name.rsplit(".", 1)[0]] if name.endswith(".weight") else None
    return L.pack(ref) if L is not None else ref


def test_forward_backward_against_autograd():
    B, H, W = 4, 128, 96
    net, state, img, joints, mask = make_case(B, H, W, seed=3)
    hm = net.forward_train(torch.from_numpy(img).cuda())
    loss, pj, grad, wgt = ops.mse_fwd_bwd(hm, joints=torch.from_numpy(joints).cuda(),
                                          mask=torch.from_numpy(mask).cuda(), stride=4.0,
                                          sigma=2.0, radius=6)
    net.backward({"final": grad})
    torch.cuda.synchronize()
    assert not net.tape

    t, w = oracle.render_targets(joints, H // 4, W // 4, 4.0, 2.0, 6)
    rl, rhm, rg = to.hrnet_train_reference(net.spec, state, img, t, w, mask,
                                           pnn.IMAGENET_MEAN, pnn.IMAGENET_STD)
    fl, fhm, fg = to.hrnet_train_reference(net.spec, state, img, t, w, mask,
                                           pnn.IMAGENET_MEAN, pnn.IMAGENET_STD, round_bf16=False)
    e_hm, s_hm = rel_l2(hm.cpu(), rhm), rel_l2(fhm, rhm)
    print(f"heatmaps: gpu-vs-bf16ref {e_hm:.3e}, f32ref-vs-bf16ref {s_hm:.3e}")
    print(f"loss: gpu {loss.item():.6e} bf16ref {rl:.6e} f32ref {fl:.6e}")
    assert e_hm <= max(2e-2, 3 * s_hm)
    assert abs(loss.item() - rl) <= max(2e-2, 3 * abs(fl - rl) / abs(rl)) * abs(rl)
    worst = 0.0
    for L in net.layers:
        names = [L.name + ".weight"] + ([L.name + ".bias"] if L.kind == "head"
                                        else [L.name + ".bn.gamma", L.name + ".bn.beta"])
        for nm in names:
            got = net.view(net.grads, nm).cpu()
            ref, f32 = packed(net, nm, rg[nm]), packed(net, nm, fg[nm])
            ref, f32 = ref.reshape(got.shape), f32.reshape(got.shape)
            e, s = rel_l2(got, ref), rel_l2(f32, ref)
            worst = max(worst, e)
            print(f"{nm:36s} gpu-vs-bf16ref {e:.3e}   f32ref-vs-bf16ref {s:.3e}")
            assert e <= max(3e-2, 3 * s), nm
    assert worst > 0.0


def test_single_exchange_modules_against_autograd():
    """Each exchange module on its own, fed the same (bf16) branch inputs and output gradients
    as the CPU graph.  At most four batch norms lie between a parameter and the seeds, so ReLU
    masks rarely differ between the two and the wiring of the hand-written backward (gradient
    accumulation over several consumers of one activation, stride-2 chains, upsampled 1x1
    projections, narrow 32-channel layers) is pinned tightly.
    Bounds: outputs within one bf16 ulp of the largest value, gradients 2e-2 relative L2."""
    from pose_estimation_amd.graph import Var

    net = hrnet.TrainablePoseHRNet("cuda:0", num_joints=17, **SMALL)
    state = hrnet.init_state(net.spec, seed=11)
    randomize_bn(state, seed=12)
    net.load_state(state)
    g = torch.Generator().manual_seed(13)
    B, H, W = 3, 32, 16
    for si, mods in enumerate(net.spec.stages):
        m = mods[0]
        nb = len(m.branches)
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
        routs, rdxs, rg = to.hrnet_module_train_reference(m, state, xs, douts)
        for o, r in zip(outs, routs):
            err = (o.t.float().cpu() - r).abs().max().item()
            assert err <= 8e-3 * r.abs().max().item() + 1e-6
        for i, (v, r) in enumerate(zip(xv, rdxs)):
            e = rel_l2(v.grad.float().cpu(), r)
            print(f"stage{si + 2} dx[{i}] rel L2 {e:.2e}")
            assert e <= 2e-2
        assert len(rg) > 0
        for nm, ref in rg.items():
            got = net.view(net.grads, nm).cpu()
            e = rel_l2(got, packed(net, nm, ref).reshape(got.shape))
            print(f"   {nm:36s} rel L2 {e:.2e}")
            assert e <= 2e-2, nm


def test_training_reduces_the_loss_and_is_reproducible():
    B, H, W = 8, 64, 64
    losses = []
    for rep in range(2):
        net, state, img, joints, mask = make_case(B, H, W, seed=7)
        it = torch.from_numpy(img).cuda()
        jt = torch.from_numpy(joints).cuda()
        run = []
        for _ in range(12):
            loss, _ = net.train_step(it, jt, lr=1e-3)
            run.append(loss.item())
        losses.append(run)
    print("losses", losses[0])
    assert losses[0][-1] < 0.7 * losses[0][0]
    assert losses[0] == losses[1]
    # structural zeros of the narrow / stem packings stay zero through the optimizer
    for name, m in net.masks.items():
        w = net.view(net.master, name + ".weight")
        assert not (w * (1 - m)).any(), name


def test_inference_mode_matches_the_inference_network():
    """The graph in inference mode (folded running statistics) against PoseHRNet on one state."""
    B, H, W = 2, 128, 96
    net, state, img, joints, mask = make_case(B, H, W, seed=17)
    ref_net = hrnet.PoseHRNet("cuda:0", num_joints=17, **SMALL)
    ref_net.load_state(state)
    it = torch.from_numpy(img).cuda()
    a, b = net.predict(it), ref_net.predict(it)
    # the two fold the running statistics on different devices: scales differ in the last place
    assert rel_l2(a.cpu(), b.cpu()) <= 1e-2


def test_full_width_step_runs_at_384x288():
    """HRNet-W32 itself, one step at the high-resolution configuration's size (batch 2)."""
    net = hrnet.TrainablePoseHRNet("cuda:0", num_joints=17)
    net.load_state(hrnet.init_state(net.spec, seed=5))
    img = torch.from_numpy(synthetic.images_u8(2, 384, 288, seed=1)).cuda()
    joints = torch.from_numpy(synthetic.joints(2, 17, 384, 288, seed=2)).cuda()
    l0, _ = net.train_step(img, joints, lr=1e-3)
    l1, _ = net.train_step(img, joints, lr=1e-3)
    assert np.isfinite(l0.item()) and np.isfinite(l1.item())
    assert torch.isfinite(net.master).all()
    hm = net.predict(img)
    assert tuple(hm.shape) == (2, 17, 96, 72)


def test_layers_are_announced_to_the_bucketer_in_descending_arena_order():
    """A branched backward finishes layers out of arena order; the gradient exchange is told
    about a layer only once every layer above it is finished, and about every layer."""
    class Recorder:
        grad_scale = 0.5

        def __init__(self):
            self.calls = []

        def layer_done(self, start):
            self.calls.append(start)

        def finish(self):
            pass

    net, state, img, joints, mask = make_case(2, 64, 64, seed=23)
    net.bucketer = Recorder()
    net.train_step(torch.from_numpy(img).cuda(), torch.from_numpy(joints).cuda(), lr=1e-3)
    assert net.bucketer.calls == sorted(net.layer_starts, reverse=True)


def test_trainer_builds_and_steps_the_hrnet_backbone():
    from pose_estimation_amd import Trainer

    tr = Trainer({"model_config": {"backbone": "hrnet_w32", "modules": [1, 1, 1], "blocks": 1,
                                   "layer1_blocks": 1, "sigma": 2.0},
                  "train_config": {"batch_size": 4, "preprocess": {"image_resize": [128, 96]}}},
                 device="cuda:0")
    tr.build_model()
    losses = []
    for feats, labels in tr.get_features_labels_data(num_batches=3):
        losses.append(tr.train_step(feats, labels).item())
    assert all(np.isfinite(losses))
    hm = tr.predict(feats["images"])
    xy, conf, idx = tr.decode(hm)
    assert tuple(hm.shape) == (4, 17, 32, 24) and tuple(xy.shape) == (4, 17, 2)


def test_hip_graph_replay_repeats_the_eager_steps_bit_for_bit():
    """The captured forward/loss/backward pass replays the same kernels in the same order: the
    loss trajectory and the parameters equal the eager run's exactly; running statistics too
    (the warm-up pass before capture leaves them untouched)."""
    B, H, W = 4, 64, 64
    runs = []
    for use_graph in (False, True):
        net, state, img, joints, mask = make_case(B, H, W, seed=29)
        jt = torch.from_numpy(joints).cuda()
        mt = torch.from_numpy(mask).cuda()
        losses = []
        for i in range(5):
            it = torch.from_numpy(synthetic.images_u8(B, H, W, seed=40 + i)).cuda()   # new data every step
            loss, _ = net.train_step(it, jt, mt, lr=1e-3, hip_graph=use_graph)
            losses.append(loss.item())
        runs.append((losses, net.master.clone(), net.bn_mean["stem.conv2"].clone()))
    assert len(net._step_graphs) == 1
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1])
    assert torch.equal(runs[0][2], runs[1][2])
