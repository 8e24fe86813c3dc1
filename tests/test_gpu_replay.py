"""Launch tape (pe_tape_*) and the host-side replay of a recorded training pass.

The tape stores the library's launches with their arguments; a replay must enqueue exactly the
recorded kernels on the recorded buffers, so the checks are bit for bit: a replayed kernel
against the same entry point called directly, and replayed training steps (new images every
step) against the eager steps of an identically initialised network - loss trajectory,
parameters and running statistics.  Nothing in the reference pins this (its session loop is
absent); the eager path is the oracle-checked one.
"""
import ctypes

import numpy as np
import pytest
import torch

from pose_estimation_amd import _ffi, hrnet, mobilenet as mb, ops, replay, synthetic
from pose_estimation_amd import model as pm
from pose_estimation_amd.training import TrainablePoseResNet

pytestmark = pytest.mark.gpu


def test_tape_replays_recorded_launches_through_the_c_abi():
    lib = _ffi.lib()
    B, K, H, W = 8, 17, 64, 48
    hm = torch.from_numpy(synthetic.decode_heatmaps(B, K, H, W, seed=3)).cuda()
    want = ops.decode(hm)
    torch.cuda.synchronize()
    # record one decode, then replay it on new contents of the same buffer
    tape = ctypes.c_void_p()
    assert lib.pe_tape_length() == -1
    _ffi.check(lib.pe_tape_begin())
    assert lib.pe_tape_begin() == _ffi.PE_E_INVALID          # one tape per thread
    got = ops.decode(hm)
    n = lib.pe_tape_length()
    assert n >= 1
    _ffi.check(lib.pe_tape_end(ctypes.byref(tape)))
    assert lib.pe_tape_length() == -1 and lib.pe_tape_end(ctypes.byref(tape)) == _ffi.PE_E_INVALID
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    hm2 = torch.from_numpy(synthetic.decode_heatmaps(B, K, H, W, seed=4)).cuda()
    want2 = [t.clone() for t in ops.decode(hm2)]
    hm.copy_(hm2)
    for t in got:
        t.zero_()
    _ffi.check(lib.pe_tape_replay(tape, 0, n))
    torch.cuda.synchronize()
    for a, b in zip(got, want2):
        assert torch.equal(a, b)
    assert lib.pe_tape_replay(tape, 0, n + 1) == _ffi.PE_E_INVALID
    assert lib.pe_tape_replay(None, 0, 0) == _ffi.PE_E_INVALID
    _ffi.check(lib.pe_tape_destroy(tape))


def test_recorded_pass_rejects_what_it_cannot_repeat():
    x = torch.ones(1024, device="cuda")

    def allocating(t):
        return t + 1.0            # a fresh result every call: its address cannot be replayed

    with pytest.raises(replay.NotReplayable, match="add"):
        replay.RecordedPass(allocating, [x], torch.device("cuda", 0))
    assert _ffi.lib().pe_tape_length() == -1      # the tape was closed on the way out

    def in_place(t):
        t.add_(1.0)
        z = torch.zeros_like(t)   # replayed as a fill of the same memory
        z.add_(t)
        return z

    rp = replay.RecordedPass(in_place, [x], torch.device("cuda", 0))
    assert torch.all(rp.static_out == 2.0)
    out = rp(torch.full_like(x, 5.0))
    assert torch.all(out == 6.0) and rp.torch_ops == 3 and rp.launches == 0
    side = torch.cuda.Stream()
    with torch.cuda.stream(side), pytest.raises(RuntimeError, match="stream it was recorded on"):
        rp(x)


def _hrnet(seed):
    net = hrnet.TrainablePoseHRNet("cuda:0", num_joints=17, sigma=2.0, width=32,
                                   modules=(1, 1, 1), blocks=1, layer1_blocks=1)
    state = hrnet.init_state(net.spec, seed=seed)
    pm.randomize_bn(state, seed=seed + 1)
    state["final.weight"] = state["final.weight"] * 100.0
    net.load_state(state)
    return net, "stem.conv2"


def _mobilenet(seed):
    net = mb.TrainablePoseMobileNetV2("cuda:0", num_joints=17, pairs=mb.COCO_PAIRS[:5],
                                      setting=mb.MBV2_SETTING[:4])
    state = mb.init_state(net.blocks, 17, 5, seed=seed)
    pm.randomize_bn(state, seed=seed + 1)
    net.load_state(state)
    return net, net.layers[0].name


def _resnet(seed):
    net = TrainablePoseResNet("cuda:0", num_joints=17, layers=(1, 1, 1, 1))
    state = pm.init_state(net.specs, seed=seed)
    pm.randomize_bn(state, seed=seed + 1)
    state["final.weight"] = state["final.weight"] * 100.0
    net.load_state(state)
    return net, net.specs[0].name


@pytest.mark.parametrize("make, optimizer", [(_hrnet, "adam"), (_mobilenet, "sgd"),
                                             (_resnet, "adam")])
def test_host_replay_repeats_the_eager_steps_bit_for_bit(make, optimizer):
    B, H, W = 4, 64, 64
    runs = []
    for use_replay in (False, True):
        net, bn_name = make(seed=29)
        jt = torch.from_numpy(synthetic.joints(B, 17, H, W, seed=2)).cuda()
        hs = int(getattr(net, "heatmap_stride", None) or net.output_stride)
        mt = torch.from_numpy(synthetic.ignore_mask(B, H // hs, W // hs, seed=4)).cuda()
        losses = []
        for i in range(5):
            it = torch.from_numpy(synthetic.images_u8(B, H, W, seed=40 + i)).cuda()   # new data
            loss, per_joint = net.train_step(it, jt, mt, lr=1e-3, optimizer=optimizer,
                                             host_replay=use_replay)
            losses.append((loss.item(), per_joint.cpu().numpy().tolist()))
        torch.cuda.synchronize()
        runs.append((losses, net.master.clone(), net.bn_mean[bn_name].clone(),
                     net.grads.clone()))
    assert len(net._recorded) == 1
    rp = next(iter(net._recorded.values()))
    print(f"{make.__name__}: {rp.launches} launches, {rp.torch_ops} torch ops per pass")
    assert rp.launches > 50
    assert np.isfinite([l for l, _ in runs[0][0]]).all()
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1:], runs[1][1:]):
        assert torch.equal(a, b)


def test_recorded_pass_survives_growth_of_the_shared_scratch_buffers():
    """The tape holds the ADDRESS of the module-level weight-gradient workspace.  A second,
    larger network makes that workspace grow, which frees the old buffer; the recording must
    keep the buffer it recorded alive, or its later replays write split partial sums into
    memory somebody else owns.  Net A replays two steps, net B (deeper: larger batched
    weight-gradient jobs) trains eagerly and the freed memory is handed out and overwritten,
    then A replays three more steps: bit for bit what five eager steps of A give."""
    from pose_estimation_amd import nn as pnn

    B, H, W = 4, 64, 64
    jt = torch.from_numpy(synthetic.joints(B, 17, H, W, seed=2)).cuda()
    mt = torch.from_numpy(synthetic.ignore_mask(B, H // 4, W // 4, seed=4)).cuda()
    imgs = [torch.from_numpy(synthetic.images_u8(B, H, W, seed=60 + i)).cuda() for i in range(5)]

    def steps(net, rng, use_replay):
        out = []
        for i in rng:
            loss, _ = net.train_step(imgs[i], jt, mt, lr=1e-3, host_replay=use_replay)
            out.append(loss.item())
        return out

    eager, _ = _resnet(seed=51)
    want = steps(eager, range(5), False)
    want_master = eager.master.clone()
    del eager
    pnn._wgrad_ws.clear()           # net A records against a workspace sized for net A alone
    torch.cuda.empty_cache()

    a, _ = _resnet(seed=51)
    got = steps(a, range(2), True)
    rp = next(iter(a._recorded.values()))
    recorded_ws = [t for t in rp.keep if any(t is v for v in pnn._wgrad_ws.values())]
    assert recorded_ws, "the recording does not hold the weight-gradient workspace"
    old_ptr, old_size = recorded_ws[0].data_ptr(), recorded_ws[0].numel()

    big = TrainablePoseResNet("cuda:0", num_joints=17, layers=(3, 4, 6, 3))
    big.load_state(pm.init_state(big.specs, seed=7))
    steps(big, range(2), False)
    torch.cuda.synchronize()
    now = pnn._wgrad_ws[pnn._dev_key("cuda:0")]
    assert now.numel() > old_size and now.data_ptr() != old_ptr, "the workspace did not grow"
    # whatever the allocator has free is handed out and overwritten
    junk = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 18, 1 << 20, 1 << 22, 1 << 24)]
    torch.cuda.synchronize()
    assert all(j.data_ptr() != old_ptr for j in junk)      # the recorded buffer was not re-issued
    got += steps(a, range(2, 5), True)
    torch.cuda.synchronize()
    assert got == want
    assert torch.equal(a.master, want_master)


def test_host_replay_is_exclusive_with_the_gradient_exchange():
    net, _ = _hrnet(seed=3)
    net.bucketer = object()
    with pytest.raises(RuntimeError, match="exclusive"):
        net.replay_host(net._pass_u8, None, None, None)
