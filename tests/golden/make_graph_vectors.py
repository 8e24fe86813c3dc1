"""Freeze the CPU oracle's outputs for the tape-graph networks as
tests/golden/graph_vectors.npz: one HRNet-style exchange module (stage 3: three branches,
strided chains, upsampled projections, the narrow 32-channel branch) and two MobileNetV2
inverted-residual blocks (stride 1 with a residual, stride 2), forward with batch statistics
and autograd backward.

    python tests/golden/make_graph_vectors.py

As with oracle_vectors.npz these pin the ORACLE (the reference holds no implementation):
tests/test_golden_vectors.py replays them against the oracle (CPU) and libpe_hip.so (GPU).
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import torch_oracle as to  # noqa: E402
from pose_estimation_amd import hrnet  # noqa: E402
from pose_estimation_amd import mobilenet as mb  # noqa: E402
from pose_estimation_amd.model import randomize_bn  # noqa: E402

OUT = Path(__file__).resolve().parent / "graph_vectors.npz"
HR_CFG = dict(width=32, modules=(1, 1, 1), blocks=1, layer1_blocks=1)
HR_STAGE = 1                     # spec.stages[1] = "stage3": branches of 32 / 64 / 128 channels
HR_B, HR_H, HR_W = 3, 32, 16        # 96 cells per channel on the narrowest branch: batch statistics
                                     # over fewer make BN backward amplify summation order past 2e-2
MB_SETTING = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 2, 2)]
MB_BLOCKS = (2, 3)               # block2: stride 1 + residual, block3: stride 2
MB_B, MB_H, MB_W = 2, 12, 10
SAMPLE_STEP = 53


def hrnet_case():
    spec = hrnet.hrnet_specs(17, **HR_CFG)
    state = hrnet.init_state(spec, seed=41)
    randomize_bn(state, seed=42)
    m = spec.stages[HR_STAGE][0]
    g = torch.Generator().manual_seed(43)
    nb = len(m.branches)
    xs = [to.bf16r(torch.randn((HR_B, HR_H >> i, HR_W >> i, spec.widths[i]), generator=g))
          for i in range(nb)]
    douts = [to.bf16r(torch.randn((HR_B, HR_H >> i, HR_W >> i, spec.widths[i]), generator=g))
             for i in range(len(m.fuse))]
    return spec, state, m, xs, douts


def mobilenet_case():
    blocks = mb.mobilenet_v2_specs(8, MB_SETTING)
    state = mb.init_state(blocks, 17, 0, seed=51)
    randomize_bn(state, seed=52)
    g = torch.Generator().manual_seed(53)
    cases = []
    for bi in MB_BLOCKS:
        b = blocks[bi]
        x = to.bf16r(torch.randn((MB_B, MB_H, MB_W, b.cin), generator=g))
        ho, wo = (MB_H - 1) // b.stride + 1, (MB_W - 1) // b.stride + 1
        dout = to.bf16r(torch.randn((MB_B, ho, wo, b.cout), generator=g))
        cases.append((b, x, dout))
    return blocks, state, cases


def main():
    v = {}
    spec, state, m, xs, douts = hrnet_case()
    outs, dxs, rg = to.hrnet_module_train_reference(m, state, xs, douts)
    for i, (o, d) in enumerate(zip(outs, dxs)):
        v[f"hr_out{i}"] = o.numpy()
        v[f"hr_dx{i}"] = d.numpy()
    for nm in sorted(rg):
        flat = rg[nm].reshape(-1)
        v["hr_g_" + nm] = flat[::SAMPLE_STEP].numpy() if nm.endswith(".weight") else flat.numpy()
        v["hr_n_" + nm] = np.float32(flat.norm().item())
    blocks, mstate, cases = mobilenet_case()
    for b, x, dout in cases:
        out, dx, g = to.mobilenet_block_train_reference(b, mstate, x, dout)
        v[f"mb_{b.name}_out"] = out.numpy()
        v[f"mb_{b.name}_dx"] = dx.numpy()
        for nm in sorted(g):
            v["mb_g_" + nm] = g[nm].reshape(-1).numpy()
    np.savez_compressed(OUT, **v)
    print(f"wrote {OUT} ({OUT.stat().st_size / 1024:.0f} KiB, {len(v)} arrays)")


if __name__ == "__main__":
    main()
