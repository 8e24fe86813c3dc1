"""Freeze the CPU oracle's outputs on seeded inputs as tests/golden/oracle_vectors.npz.

    python tests/golden/make_oracle_vectors.py

The reference holds no implementation to generate vectors from (its source files are absent), so
these vectors pin the ORACLE, not the reference: they keep the restatement from drifting between
rounds and give the HIP path fixed numbers to hit.  tests/test_golden_vectors.py replays them
against the oracle (CPU) and against libpe_hip.so (GPU).
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import oracle  # noqa: E402
from oracle import torch_oracle as to  # noqa: E402
from pose_estimation_amd import model as pm  # noqa: E402
from pose_estimation_amd import synthetic  # noqa: E402

OUT = Path(__file__).resolve().parent / "oracle_vectors.npz"

BLOCK_SHAPE = (2, 8, 6, 256)     # NHWC input of layer1's second bottleneck (identity branch)
DECONV_SHAPE = (1, 4, 3, 256)    # NHWC input of the second deconv (256 -> 256)
WEIGHT_SEED = 31
SAMPLE_STEP = 97                 # weight gradients are stored at every 97th packed element
VEC_PAIRS = [(0, 1), (2, 1), (5, 7), (11, 13), (0, 2)]   # (0, 2) pairs two off-map centres


def adversarial_maps(H=64, W=48):
    """Heat maps that exercise ties, borders, non-positive peaks and plateaus."""
    maps = np.zeros((8, H, W), np.float32)
    maps[1] = -1.0                                   # all negative: conf <= 0 zeroes coordinates
    maps[2, 10, 7] = maps[2, 40, 30] = 0.75          # exact tie: lowest flat index wins
    maps[3, 0, 0] = 1.0                              # corner: no refinement
    maps[4, H - 1, W - 1] = 1.0                      # last element
    maps[5, 20, 1] = 0.5                             # x == 1: refinement needs 1 < x
    maps[5, 20, 2] = 0.25
    maps[6, 30:33, 20:23] = 0.9                      # plateau: first element, zero gradient sign
    maps[7, 33, 25] = 0.6                            # interior peak with asymmetric neighbours
    maps[7, 33, 26] = 0.3
    maps[7, 32, 25] = 0.2
    maps[7, 34, 25] = 0.4
    return maps


def block_case():
    stem, blocks, deconvs, final = pm.resnet50_pose_specs(17, (2, 1, 1, 1))
    specs = (stem, blocks, deconvs, final)
    state = pm.init_state(specs, seed=WEIGHT_SEED)
    pm.randomize_bn(state, seed=WEIGHT_SEED + 1)
    for k in state:
        if k.startswith("deconv") and k.endswith(".weight"):
            state[k] = state[k] * 30.0
    g = torch.Generator().manual_seed(17)
    bx = to.bf16r(torch.randn(BLOCK_SHAPE, generator=g))
    bdout = to.bf16r(torch.randn(BLOCK_SHAPE, generator=g))
    dx_in = to.bf16r(torch.randn(DECONV_SHAPE, generator=g))
    B, H, W, _ = DECONV_SHAPE
    dda = to.bf16r(torch.randn((B, 2 * H, 2 * W, 256), generator=g))
    return specs, state, bx, bdout, dx_in, dda


def main():
    v = {}
    # ---- a1 decode
    hm = synthetic.decode_heatmaps(4, 17, 64, 48, seed=0)
    adv = adversarial_maps()
    hm[3, :8] = adv
    xy, conf, idx = oracle.decode_heatmaps(hm)
    v.update(decode_hm=hm, decode_xy=xy, decode_conf=conf, decode_idx=idx.astype(np.int32))
    # ---- a2 render
    joints = synthetic.joints(4, 17, 256, 192, seed=2)
    joints[0, 0] = (-30.0, 10.0, 2)       # centre far off the map: weight 0
    joints[0, 1] = (1.0, 253.0, 2)        # centre in the corner: window clipped at two borders
    joints[0, 2] = (192.0, 100.0, 2)      # centre one cell past the right edge: weight 0
    t, w = oracle.render_targets(joints, 64, 48, 4.0, 2.0, 6)
    v.update(render_joints=joints, render_target=t, render_weight=w)
    # ---- vector maps: labels for the first two images, loss + grad on noisy predictions
    vec, sup = oracle.render_vecmaps(joints[:2], VEC_PAIRS, 64, 48, 4.0, 2)
    vpred = (vec + np.random.Generator(np.random.PCG64(6)).normal(0, 0.5, vec.shape)).astype(np.float32)
    vl, vpc, vg = oracle.vecmap_loss_fwd_bwd(vpred, joints[:2], VEC_PAIRS, 4.0, 2,
                                             synthetic.ignore_mask(2, 64, 48, seed=4))
    v.update(vec_label=vec, vec_pred=vpred, vec_loss=np.float32(vl), vec_per_channel=vpc,
             vec_grad=vg)
    # ---- a3 loss + grad
    rng = np.random.Generator(np.random.PCG64(5))
    pred = (t[:2] + rng.normal(0, 0.1, t[:2].shape)).astype(np.float32)
    mask = synthetic.ignore_mask(2, 64, 48, seed=4)
    loss, pj, grad = oracle.mse_fwd_bwd(pred, t[:2], w[:2], mask)
    v.update(mse_pred=pred, mse_mask=mask, mse_loss=np.float32(loss), mse_per_joint=pj,
             mse_grad=grad)
    # ---- a6 three Adam steps on a 1 K-element arena (TF-1 form), then one SGD step
    p = rng.normal(0, 1, 1000).astype(np.float32)
    m = np.zeros(1000, np.float32)
    vv = np.zeros(1000, np.float32)
    gs = rng.normal(0, 0.1, (3, 1000)).astype(np.float32)
    v.update(adam_p0=p, adam_g=gs)
    for step in range(3):
        p, m, vv, pb = oracle.adam_step(p, m, vv, gs[step], 1e-3, 0.9, 0.999, 1e-8, step + 1)
    v.update(adam_p3=p, adam_m3=m, adam_v3=vv, adam_pb3=pb)
    sp, smom, spb = oracle.sgd_step(v["adam_p0"], np.zeros(1000, np.float32), gs[0], 0.01, 0.9)
    v.update(sgd_p1=sp, sgd_mom1=smom)
    # ---- a4 one bottleneck, a5 one deconv: forward with batch statistics and backward
    specs, state, bx, bdout, dx_in, dda = block_case()
    blk = specs[1][1]
    out, dx, rg = to.bottleneck_train_reference(blk, state, bx, bdout)
    v.update(block_out=out.numpy(), block_dx=dx.numpy())
    for c in (blk.c1, blk.c2, blk.c3):
        v[f"block_{c.name}_dgamma"] = rg[c.name + ".bn.gamma"].numpy()
        v[f"block_{c.name}_dbeta"] = rg[c.name + ".bn.beta"].numpy()
        packed = pm.pack_weight(c, rg[c.name + ".weight"]).reshape(-1)
        v[f"block_{c.name}_dw_sample"] = packed[::SAMPLE_STEP].numpy()
        v[f"block_{c.name}_dw_norm"] = np.float32(packed.norm().item())
    d = specs[2][1]
    a, ddx, dg = to.deconv_train_reference(d, state, dx_in, dda)
    packed = pm.pack_weight(d, dg[d.name + ".weight"]).reshape(-1)
    v.update(deconv_out=a.numpy(), deconv_dx=ddx.numpy(),
             deconv_dgamma=dg[d.name + ".bn.gamma"].numpy(),
             deconv_dbeta=dg[d.name + ".bn.beta"].numpy(),
             deconv_dw_sample=packed[::SAMPLE_STEP].numpy(),
             deconv_dw_norm=np.float32(packed.norm().item()))
    np.savez_compressed(OUT, **v)
    print(f"wrote {OUT} ({OUT.stat().st_size / 1024:.0f} KiB, {len(v)} arrays)")


if __name__ == "__main__":
    main()
