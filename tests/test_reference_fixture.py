"""The only results the reference pins for this path: outputs printed by its notebook.

tests/golden/reference_notebook_fixture.json holds the non-zero index lists of one 40x40x3
heat-map label (test.ipynb:606-614): five 3x3 blocks, and of the 40x40x8 vector-map label of
the same sample (test.ipynb:634-653): 116 entries.  The label is multi-person (two instances
share channels 0 and 1), so the instances are rendered as separate (person, joint) maps and
merged (maximum for heat maps, sum for the disjoint vector maps); radius 1 at stride 8 must
reproduce both supports exactly.  Heat-map values are not pinned by the reference beyond
"in (0, 1]" (test.ipynb:922); vector-map values are pinned through their zero pattern (an offset
component equal to 0 is missing from the non-zero list), which fixes every centre.

The vector map shows a third joint of the second person whose centre is one column past the
right edge (row 19, column 40): its vector-map window survives clipped to column 39, and the
partner's entries point at the off-map centre, while heat-map channel 2 holds nothing there.
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


# ------------------------------------------------------------------ vector maps
PAIRS = [(0, 1), (2, 1)]          # the reference's drawing cell: pairs = [[0, 1], [2, 1]]


def two_person_joints(stride):
    """Both persons of the label with all three joints; person 1's joint 2 sits at (19, 40)."""
    def px(r, c):
        return (c * stride + 1.5, r * stride - 2.5, 2)

    j = np.zeros((2, 3, 3), np.float32)
    j[0, 0], j[0, 1], j[0, 2] = px(5, 18), px(6, 15), px(12, 18)
    j[1, 0], j[1, 1], j[1, 2] = px(13, 38), px(19, 38), px(19, 40)
    return j


def assert_vec_support(label_hwc, fix):
    rows, cols, chans = label_hwc.nonzero()
    nz = fix["vecmaps_nonzero"]
    assert rows.tolist() == nz["rows"]
    assert cols.tolist() == nz["cols"]
    assert chans.tolist() == nz["channels"]


def test_oracle_vecmaps_reproduce_reference_support(fixture):
    H, W, K = fixture["heatmap_shape_hwc"]
    j = two_person_joints(fixture["stride"])
    vec, sup = oracle.render_vecmaps(j, PAIRS, H, W, fixture["stride"], radius=1)
    assert len(fixture["vecmaps_nonzero"]["rows"]) == 116
    assert_vec_support(vec.sum(axis=0).transpose(1, 2, 0), fixture)
    # the same joints give the heat-map support too: the off-map joint is labelled but not drawn
    t, w = oracle.render_targets(j, H, W, fixture["stride"], sigma=1.0, radius=1)
    assert w.tolist() == [[1, 1, 1], [1, 1, 0]]
    rows, cols, chans = t.max(axis=0).transpose(1, 2, 0).nonzero()
    assert rows.tolist() == fixture["heatmaps_nonzero"]["rows"]
    assert cols.tolist() == fixture["heatmaps_nonzero"]["cols"]
    assert chans.tolist() == fixture["heatmaps_nonzero"]["channels"]
    # offsets follow the drawing cell's reading: cell + (vec[4i], vec[4i+1]) is the partner's centre
    ys, xs = np.nonzero(sup[0, 0])
    assert set((y + int(vec[0, 1, y, x]), x + int(vec[0, 0, y, x])) for y, x in zip(ys, xs)) == {(6, 15)}


@pytest.mark.gpu
def test_hip_vecmaps_reproduce_reference_support_nhwc(fixture):
    import torch

    from pose_estimation_amd import ops

    H, W, K = fixture["heatmap_shape_hwc"]
    j = two_person_joints(fixture["stride"])
    jt = torch.from_numpy(j).cuda()
    v_nhwc = ops.render_vecmaps(jt, PAIRS, H, W, fixture["stride"], radius=1, layout="NHWC")
    assert v_nhwc.shape == (2, H, W, 8)
    assert_vec_support(v_nhwc.sum(dim=0).cpu().numpy(), fixture)
    want, _ = oracle.render_vecmaps(j, PAIRS, H, W, fixture["stride"], radius=1)
    assert np.array_equal(v_nhwc.permute(0, 3, 1, 2).cpu().numpy(), want)
    assert np.array_equal(ops.render_vecmaps(jt, PAIRS, H, W, fixture["stride"], 1).cpu().numpy(), want)
    t, w = ops.render_targets(jt, H, W, fixture["stride"], sigma=1.0, radius=1)
    assert w.cpu().tolist() == [[1, 1, 1], [1, 1, 0]]


def test_oracle_vecmap_values_match_the_notebooks_printed_offsets(fixture):
    """The drawing cell prints (vecmaps[y, x, 4i], vecmaps[y, x, 4i+1]) over the first joint's
    cells in row-major order, nine per 3x3 instance (135 value pairs in all).  Each block of
    nine must be what the oracle renders for a partner at the offset the block's centre
    names; an all-zero block is an instance whose partner joint is unlabelled."""
    H, W, K = fixture["heatmap_shape_hwc"]
    stride = fixture["stride"]
    total = 0
    for group in fixture["vecmap_printed_values"]:
        k1, k2 = group["pair"]
        i = PAIRS.index((k1, k2))
        vals = group["values"]
        assert len(vals) % 9 == 0
        for n in range(0, len(vals), 9):
            block = np.array(vals[n:n + 9], np.float32)
            ox, oy = int(block[4, 0]), int(block[4, 1])
            j = np.zeros((1, K, 3), np.float32)
            j[0, k1] = (20 * stride + 1.5, 20 * stride - 2.5, 2)
            if block.any():
                j[0, k2] = ((20 + ox) * stride + 1.5, (20 + oy) * stride - 2.5, 2)
            vec, sup = oracle.render_vecmaps(j, PAIRS, H, W, stride, radius=1)
            got = np.stack([vec[0, 4 * i, 19:22, 19:22].reshape(-1),
                            vec[0, 4 * i + 1, 19:22, 19:22].reshape(-1)], axis=1)
            assert np.array_equal(got, block), (group["pair"], n)
            total += 9
    assert total == 135
