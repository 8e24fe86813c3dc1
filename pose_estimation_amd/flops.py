"""Algorithmic multiply-accumulate counts of the pose network, from its layer list.

A stride-2 transposed 4x4 conv is counted at its true 4 taps per output pixel; the stem at its
7x7x3 taps (not the padded space-to-depth K the kernel runs).  FLOPs = 2 * MACs.
"""
from __future__ import annotations

from .model import resnet50_pose_specs


def forward_macs(in_h: int = 256, in_w: int = 192, num_joints: int = 17, layers=(3, 4, 6, 3),
                 width: int = 64, head_ch: int = 256):
    """Returns (total MACs per image, per-layer list of (name, macs), parameter count)."""
    stem, blocks, deconvs, final = resnet50_pose_specs(num_joints, layers, width, head_ch)
    rows = []
    params = 0
    h, w = in_h // 2, in_w // 2
    rows.append((stem.name, h * w * stem.cout * 49 * 3))
    params += stem.cout * 49 * 3 + 2 * stem.cout
    h, w = h // 2, w // 2  # max pool
    for b in blocks:
        s = b.c2.stride
        rows.append((b.c1.name, h * w * b.c1.cout * b.c1.cin))
        ho, wo = h // s, w // s
        rows.append((b.c2.name, ho * wo * b.c2.cout * b.c2.cin * 9))
        rows.append((b.c3.name, ho * wo * b.c3.cout * b.c3.cin))
        params += b.c1.cout * b.c1.cin + b.c2.cout * b.c2.cin * 9 + b.c3.cout * b.c3.cin
        params += 2 * (b.c1.cout + b.c2.cout + b.c3.cout)
        if b.down:
            rows.append((b.down.name, ho * wo * b.down.cout * b.down.cin))
            params += b.down.cout * b.down.cin + 2 * b.down.cout
        h, w = ho, wo
    for d in deconvs:
        h, w = 2 * h, 2 * w
        rows.append((d.name, h * w * d.cout * d.cin * 4))
        params += d.cout * d.cin * 16 + 2 * d.cout
    rows.append((final.name, h * w * final.cout * final.cin))
    params += final.cout * final.cin + final.cout
    return sum(m for _, m in rows), rows, params


if __name__ == "__main__":
    total, rows, params = forward_macs()
    for name, m in rows:
        print(f"{name:20s} {m / 1e6:10.2f} MMAC")
    print(f"total {total / 1e9:.3f} GMAC = {2 * total / 1e9:.2f} GFLOP per image, "
          f"{params / 1e6:.2f} M parameters")
