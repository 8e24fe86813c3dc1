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


def conv_train_bytes(m_in: int, cin: int, m_out: int, cout: int, w_elems: int,
                     residual: bool = False, bn: bool = True) -> int:
    """HBM bytes one conv (+ batch norm + activation) layer of a training step must move when
    every element-wise operation rides inside a GEMM and nothing is read twice by one pass:

      forward          read X, read W, write Z                       (bf16)
      backward-data    read dZ, read W, write dX                     (bf16)
      backward-weight  read X, read dZ, write dW                     (dW float32)
      batch norm       its gradient needs z once more in backward    (read Z)
      residual layer   forward reads the identity and writes the sum; backward reads the
                       gradient the other consumer left             (3 x the output)

    m_in / m_out: pixels (batch included) of the input / output maps.  Batch statistics force
    Z to be stored before it can be normalised, so normalising is priced as free work on the
    consumer's load, not as a pass of its own."""
    x, z, w = 2 * m_in * cin, 2 * m_out * cout, 2 * w_elems
    b = (x + w + z) + (z + w + x) + (x + z + 4 * w_elems)
    if bn:
        b += z
    if residual:
        b += 3 * z
    return b


def train_step_bytes(batch: int = 64, in_h: int = 256, in_w: int = 192, num_joints: int = 17,
                     layers=(3, 4, 6, 3), width: int = 64, head_ch: int = 256):
    """Algorithmic HBM bytes of one ResNet-50 pose training step (the denominator for measured
    step traffic): conv_train_bytes over the layer list, the max-pool (read the stem's z, write
    the pooled map; backward the same two tensors), the uint8 images and their space-to-depth
    copy, the float32 heat maps (written, read by the loss, gradient written and read) and the
    optimizer's 30 bytes per parameter.  Returns (total, {group: bytes})."""
    stem, blocks, deconvs, final = resnet50_pose_specs(num_joints, layers, width, head_ch)
    _, _, params = forward_macs(in_h, in_w, num_joints, layers, width, head_ch)
    g = {}
    h, w = in_h // 2, in_w // 2
    m = batch * h * w
    g["input"] = batch * in_h * in_w * 3 + 2 * m * 16
    g["stem"] = conv_train_bytes(m, 16, m, stem.cout, stem.cout * 4 * 64)
    hp, wp = h // 2, w // 2
    g["pool"] = 2 * (2 * m * stem.cout + 2 * batch * hp * wp * stem.cout)
    h, w = hp, wp
    tot = 0
    for b in blocks:
        s = b.c2.stride
        ho, wo = h // s, w // s
        mi, mo = batch * h * w, batch * ho * wo
        tot += conv_train_bytes(mi, b.c1.cin, mi, b.c1.cout, b.c1.cin * b.c1.cout)
        tot += conv_train_bytes(mi, b.c2.cin, mo, b.c2.cout, 9 * b.c2.cin * b.c2.cout)
        tot += conv_train_bytes(mo, b.c3.cin, mo, b.c3.cout, b.c3.cin * b.c3.cout, residual=True)
        if b.down:
            tot += conv_train_bytes(mi, b.down.cin, mo, b.down.cout, b.down.cin * b.down.cout)
        h, w = ho, wo
    g["blocks"] = tot
    tot = 0
    for d in deconvs:
        tot += conv_train_bytes(batch * h * w, d.cin, batch * 4 * h * w, d.cout, 16 * d.cin * d.cout)
        h, w = 2 * h, 2 * w
    g["deconvs"] = tot
    mo = batch * h * w
    g["final+loss"] = conv_train_bytes(mo, final.cin, 0, 0, final.cin * final.cout, bn=False) \
        + 4 * (4 * mo * final.cout)
    g["optimizer"] = 30 * params
    return sum(g.values()), g


if __name__ == "__main__":
    total, rows, params = forward_macs()
    for name, m in rows:
        print(f"{name:20s} {m / 1e6:10.2f} MMAC")
    print(f"total {total / 1e9:.3f} GMAC = {2 * total / 1e9:.2f} GFLOP per image, "
          f"{params / 1e6:.2f} M parameters")
    tb, groups = train_step_bytes()
    print(f"training step, batch 64: {tb / 1e9:.2f} GB algorithmic HBM bytes: "
          + ", ".join(f"{k} {v / 1e9:.2f}" for k, v in groups.items()))
