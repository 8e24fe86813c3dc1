"""Host-side logic of the tape-graph networks that needs no GPU: layer lists, packed layouts,
parameter masks, channel padding, and the narrow backward-data weight layout the library
reports (a host-only entry point of the C ABI)."""
import pytest
import torch

from pose_estimation_amd import hrnet
from pose_estimation_amd import mobilenet as mb
from pose_estimation_amd import nn as pnn
from pose_estimation_amd.graph import Layer


def test_narrow_backward_data_layout_is_a_valid_run_packing():
    """Every tap gets a (slot, sub-block) of its own; taps sharing a slot read adjacent gradient
    columns of one input-parity class and one filter row, in ascending column order."""
    for k, s, p, cout in [(3, 1, 1, 32), (3, 2, 1, 32), (1, 1, 0, 32), (3, 1, 1, 16), (3, 2, 1, 8),
                          (5, 1, 2, 16), (7, 2, 3, 8)]:
        slot, sub, n = pnn.narrow_bwd_layout(k, s, p, cout)
        P = 64 // cout
        assert len(set(zip(slot, sub))) == k * k and max(slot) == n - 1 and max(sub) < P
        for q in range(n):
            taps = [(kh, kw) for kh in range(k) for kw in range(k) if slot[kh * k + kw] == q]
            assert len({kh for kh, _ in taps}) == 1
            kh = taps[0][0]
            cols = {}
            for _, kw in taps:
                px = (kw - p) % s
                assert (px + p - kw) % s == 0
                cols[sub[kh * k + kw]] = ((px + p - kw) // s, px)
            start = cols[0][0]
            assert len({px for _, px in cols.values()}) == 1          # one parity class
            assert all(col == start + j for j, (col, _) in cols.items())
    # 3x3 stride 1 on 32 channels: 6 slots of 64 instead of 9 taps padded to 64 channels
    assert pnn.narrow_bwd_layout(3, 1, 1, 32)[2] == 6
    assert pnn.bwd_data_weight_shape(64, 32, 3, 1, 1) == (64, 6, 64)
    assert pnn.bwd_data_weight_shape(32, 64, 3, 1, 1) == (32, 9, 64)


def test_layer_packings_and_masks():
    g = torch.Generator().manual_seed(1)
    L = Layer("a", "conv", 32, 64, 3, 1, 1)
    assert L.narrow and L.packed_shape() == (64, 3, 2, 64)
    w = torch.randn(64, 32, 3, 3, generator=g)
    p = L.pack(w)
    assert p.dtype == torch.float32 and tuple(p.shape) == L.packed_shape()
    m = L.param_mask()
    assert m.sum().item() == w.numel() and torch.equal(p * m, p)
    assert torch.equal(p.to(torch.bfloat16), pnn.pack_conv_weight_narrow(w))
    assert Layer("b", "conv", 64, 64, 3).param_mask() is None
    S = Layer("s", "stem3", 3, 64, 3, 2, 2)
    assert S.packed_shape() == (64, 2, 64) and S.param_mask().sum().item() == 64 * 27
    D = Layer("d", "dw", 128, 128, 3, 2, 2, 1)
    wd = torch.randn(128, 1, 3, 3, generator=g)
    assert D.packed_shape() == (9, 128) and not D.narrow
    assert torch.equal(D.pack(wd)[4], wd[:, 0, 1, 1])
    H = Layer("h", "head", 32, 17)
    assert H.narrow and H.packed_shape() == (17, 1, 1, 64) and H.kpad == 64
    assert Layer("v", "head", 320, 76).kpad == 128


def test_hrnet_graph_layers_follow_execution_order():
    spec = hrnet.hrnet_specs()
    layers = hrnet._graph_layers(spec)
    assert [L.name for L in layers[:-1]] == [u.name for u in hrnet.all_units(spec)]
    assert layers[0].kind == "stem3" and layers[-1].kind == "head" and layers[-1].cin == 32
    assert sum(L.narrow for L in layers) == 90
    # every stride-2 chain out of the 32-channel branch starts with a narrow-input conv
    assert all(L.narrow for L in layers if ".fuse" in L.name and L.cin == 32)


def test_mobilenet_channel_padding():
    """Channel counts stay as the architecture gives them (all multiples of 8 in MobileNetV2-1.0);
    only a count that is not a whole number of 16-byte chunks is padded, with zero weights and
    switched-off norms."""
    blocks = mb.mobilenet_v2_specs()
    layers = mb._graph_layers(blocks, 17, 19)
    by = {L.name: L for L in layers}
    assert (by["block1.expand"].cin, by["block1.expand"].cout) == (16, 96)
    assert (by["block1.project"].cin, by["block1.project"].cout) == (96, 24)
    assert by["block16.project"].cout == 320 and by["stem"].cout == 32
    assert by["block1.expand"].narrow and not by["block2.expand"].narrow     # 16 vs 24 channels
    assert [L.name for L in layers[-2:]] == ["heat", "vec"] and layers[-1].cout == 76
    st = mb.init_state(blocks, 17, 19, seed=3)
    ps = mb.pad_state(layers, st)
    assert all(torch.equal(ps[k], st[k]) for k in st if k.endswith(".weight"))
    # a width that needs padding: 12 -> 16 and 20 -> 24 channels
    blocks = mb.mobilenet_v2_specs(8, [(1, 12, 1, 1), (6, 20, 1, 2)])
    layers = mb._graph_layers(blocks, 17, 0)
    assert all(L.cout % 8 == 0 and L.cin % 8 == 0 for L in layers if L.kind not in ("head", "stem3"))
    st = mb.init_state(blocks, 17, 0, seed=3)
    ps = mb.pad_state(layers, st)
    w, wp = st["block1.project.weight"], ps["block1.project.weight"]
    assert tuple(w.shape) == (20, 72, 1, 1) and tuple(wp.shape) == (24, 72, 1, 1)
    assert torch.equal(wp[:20], w) and not wp[20:].any()
    assert ps["block1.project.bn.gamma"][20:].abs().sum() == 0     # padded norms are switched off
    w, wp = st["block1.expand.weight"], ps["block1.expand.weight"]
    assert tuple(wp.shape) == (72, 16, 1, 1) and torch.equal(wp[:, :12], w) and not wp[:, 12:].any()
    with pytest.raises(KeyError):
        mb.pad_state(layers, {})


def test_box_head_layout_follows_the_reference_anchor_order():
    """Anchors are flattened level by level as [rows, cols, 3] (1500 = 20*20*3 + 10*10*3,
    test.ipynb:423-464); the box heads' NCHW maps are re-ordered to that list and gradients
    take the inverse route."""
    from pose_estimation_amd import mobilenet as mb

    g = torch.Generator().manual_seed(0)
    maps = [torch.randn((2, 15, 20, 20), generator=g), torch.randn((2, 15, 10, 10), generator=g)]
    logit, reg = mb.box_outputs(maps)
    assert tuple(logit.shape) == (2, 1500) and tuple(reg.shape) == (2, 1500, 4)
    r, c, a = 7, 13, 2
    n = (r * 20 + c) * 3 + a
    assert logit[1, n] == maps[0][1, a * 5, r, c]
    assert torch.equal(reg[1, n], maps[0][1, a * 5 + 1:a * 5 + 5, r, c])
    n2 = 1200 + (4 * 10 + 9) * 3 + 1
    assert logit[0, n2] == maps[1][0, 5, 4, 9]
    back = mb.box_grads(logit, reg, [(20, 20), (10, 10)])
    assert all(torch.equal(b, m) for b, m in zip(back, maps))
    layers = mb._graph_layers(mb.mobilenet_v2_specs(), 17, 2,
                              det_blocks=mb.detection_specs(320))
    names = [L.name for L in layers]
    assert names[-8:] == ["det1.expand", "det1.dw", "det1.project", "box1",
                          "det2.expand", "det2.dw", "det2.project", "box2"]
    assert layers[-1].cout == 15 and layers[-1].kind == "head"


def test_oracle_detection_branch_autograd_is_consistent():
    """The CPU restatement of the detection branch: every parameter of the branch receives a
    gradient, and the trunk's gradients change when the box heads are seeded."""
    import numpy as np
    from oracle import torch_oracle as to
    from pose_estimation_amd import mobilenet as mb
    from pose_estimation_amd import nn as pnn
    from pose_estimation_amd import synthetic

    setting = [(1, 16, 1, 1), (6, 24, 1, 2)]
    blocks = mb.mobilenet_v2_specs(8, setting)
    det = mb.detection_specs(blocks[-1].cout, channels=32)
    state = mb.init_state(blocks, 3, 0, seed=1, det_blocks=det)
    img = synthetic.images_u8(2, 32, 32, seed=1)
    dheat = np.zeros((2, 3, 8, 8), np.float32)
    g = torch.Generator().manual_seed(2)
    dboxes = [torch.randn((2, 15, 4, 4), generator=g), torch.randn((2, 15, 2, 2), generator=g)]
    heat, vec, grads, maps = to.mobilenet_train_reference(
        blocks, state, img, pnn.IMAGENET_MEAN, pnn.IMAGENET_STD, dheat, det_blocks=det,
        dboxes=dboxes)
    assert vec is None and [tuple(m.shape) for m in maps] == [(2, 15, 4, 4), (2, 15, 2, 2)]
    for k, v in grads.items():
        if k.startswith(("det", "box")):
            assert v is not None and torch.isfinite(v).all() and v.abs().sum() > 0, k
    assert grads["stem.weight"].abs().sum() > 0      # reached through the detection branch only
    h2, _, g2 = to.mobilenet_train_reference(blocks, state, img, pnn.IMAGENET_MEAN,
                                             pnn.IMAGENET_STD, dheat)
    assert torch.equal(h2, heat) and g2["stem.weight"].abs().sum() == 0
