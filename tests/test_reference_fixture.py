"""The only results the reference pins for this path: outputs printed by its notebook.

tests/golden/reference_notebook_fixture.json holds the non-zero index lists of one 40x40x3
heat-map label (test.ipynb:606-614): five 3x3 blocks.  The label is multi-person (two
instances share channels 0 and 1), so the instances are rendered as separate (person, joint)
maps and merged with an element-wise maximum; radius 1 at stride 8 must reproduce the support
exactly.  Values are not pinned by the reference beyond "in (0, 1]" (test.ipynb:922).
"""
import json

import numpy as np
import pytest

import oracle


@pytest.fixture(scope="module")
def fixture(golden_dir):
    return json.loads((golden_dir / "reference_notebook_fixture.json").read_text())


def block_centres(fix):
    """Recover the 3x3 block centres (row, col, channel) from the support itself."""
    nz = fix["heatmaps_nonzero"]
    cells = set(zip(nz["rows"], nz["cols"], nz["channels"]))
    centres = []
    for (r, c, ch) in sorted(cells):
        if all((r + dr, c + dc, ch) in cells for dr in (-1, 0, 1) for dc in (-1, 0, 1)):
            centres.append((r, c, ch))
    return cells, centres


def person_joints(centres, stride, K):
    """Pack the instances into persons x K joints (x, y, v), one instance per (person, k)."""
    per_channel = {}
    for (r, c, ch) in centres:
        per_channel.setdefault(ch, []).append((r, c))
    P = max(len(v) for v in per_channel.values())
    j = np.zeros((P, K, 3), np.float32)
    for ch, lst in per_channel.items():
        for p, (r, c) in enumerate(lst):
            # any pixel whose cell rounds to (r, c); +1.5 keeps it off the rounding boundary
            j[p, ch] = (c * stride + 1.5, r * stride - 2.5, 2)
    return j


def test_fixture_is_five_3x3_blocks(fixture):
    cells, centres = block_centres(fixture)
    assert len(cells) == 45
    assert centres == [(5, 18, 0), (6, 15, 1), (12, 18, 2), (13, 38, 0), (19, 38, 1)]
    assert fixture["stride"] * fixture["heatmap_shape_hwc"][0] == fixture["input_hw"][0]


def test_oracle_render_reproduces_reference_support(fixture):
    H, W, K = fixture["heatmap_shape_hwc"]
    cells, centres = block_centres(fixture)
    j = person_joints(centres, fixture["stride"], K)
    t, w = oracle.render_targets(j, H, W, fixture["stride"], sigma=1.0, radius=1)
    label = t.max(axis=0).transpose(1, 2, 0)  # merge persons, to NHWC [H,W,K]
    rows, cols, chans = label.nonzero()
    assert set(zip(rows.tolist(), cols.tolist(), chans.tolist())) == cells
    assert rows.tolist() == fixture["heatmaps_nonzero"]["rows"]
    assert cols.tolist() == fixture["heatmaps_nonzero"]["cols"]
    assert chans.tolist() == fixture["heatmaps_nonzero"]["channels"]
    assert label.max() == 1.0 and label[label > 0].min() > 0
    assert int(w.sum()) == 5


def test_oracle_decode_finds_the_reference_block_centres(fixture):
    H, W, K = fixture["heatmap_shape_hwc"]
    _, centres = block_centres(fixture)
    j = person_joints(centres, fixture["stride"], K)
    t, w = oracle.render_targets(j, H, W, fixture["stride"], sigma=1.0, radius=1)
    xy, conf, idx = oracle.decode_heatmaps(t)
    for p in range(j.shape[0]):
        for k in range(K):
            if w[p, k]:
                r, c = [(r, c) for (r, c, ch) in centres if ch == k][p]
                assert idx[p, k] == r * W + c and conf[p, k] == 1.0
                assert xy[p, k].tolist() == [c, r]  # symmetric block: no quarter-pixel move


@pytest.mark.gpu
def test_hip_render_reproduces_reference_support_nhwc(fixture):
    """Same known answer through the C ABI, written directly in the reference's NHWC layout."""
    import torch

    import pose_estimation_amd as pe

    H, W, K = fixture["heatmap_shape_hwc"]
    cells, centres = block_centres(fixture)
    j = person_joints(centres, fixture["stride"], K)
    jt = torch.from_numpy(j).cuda()
    t_nhwc, w = pe.render_targets(jt, H, W, fixture["stride"], sigma=1.0, radius=1, layout="NHWC")
    assert t_nhwc.shape == (j.shape[0], H, W, K)
    label = t_nhwc.amax(dim=0).cpu().numpy()
    rows, cols, chans = label.nonzero()
    assert rows.tolist() == fixture["heatmaps_nonzero"]["rows"]
    assert cols.tolist() == fixture["heatmaps_nonzero"]["cols"]
    assert chans.tolist() == fixture["heatmaps_nonzero"]["channels"]
    want, _ = oracle.render_targets(j, H, W, fixture["stride"], sigma=1.0, radius=1)
    assert np.array_equal(t_nhwc.permute(0, 3, 1, 2).cpu().numpy(), want)
    # and the NCHW fast path
    t_nchw, _ = pe.render_targets(jt, H, W, fixture["stride"], sigma=1.0, radius=1)
    assert np.array_equal(t_nchw.cpu().numpy(), want)
    xy, conf, idx = pe.decode(t_nhwc, layout="NHWC")
    oxy, oconf, oidx = oracle.decode_heatmaps(want)
    assert np.array_equal(idx.cpu().numpy(), oidx)
    assert np.array_equal(xy.cpu().numpy(), oxy)
