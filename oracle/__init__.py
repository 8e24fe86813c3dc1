"""CPU oracle for the pose-estimation hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under this package is imported by the product package
(`pose_estimation_amd`).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker.

PARITY UNPINNED: the reference snapshot holds no implementation of decode,
target rendering, loss, model or optimizer (every .py named by
PoseEstimation.pyproj:22-40 is absent), so these functions restate the
top-down "simple baselines" formulation rather than reference source.  The
only reference-held facts they are pinned against are notebook outputs:
  * heat-map support of test.ipynb:606-614 (five 3x3 blocks on a 40x40x3 map),
  * vector-map support of test.ipynb:634-653 (116 non-zero cells of the 40x40x8 map
    of the same label, including a keypoint whose centre is off the map),
  * 1500 = 20*20*3 + 10*10*3 anchors, one positive anchor per person, corner-box `regs`
    (test.ipynb:423-572),
  * stride 8 / 320x320 -> 40x40 (test.ipynb:919, 928, 1010),
  * Adam hyper-parameters (test.ipynb:1001-1008).
"""
from .numpy_oracle import (  # noqa: F401
    decode_heatmaps,
    decode_heatmaps_nan_skipped,
    gaussian_table,
    render_targets,
    render_vecmaps,
    augment_u8,
    augment_mask,
    transform_joints,
    vecmap_loss_fwd_bwd,
    ssd_anchors,
    box_iou,
    boxes_from_joints,
    anchor_assign,
    box_loss_fwd_bwd,
    mse_fwd_bwd,
    adam_step,
    sgd_step,
    bf16_round,
    f32_to_bf16_bits,
    bf16_bits_to_f32,
)
