#!/usr/bin/env python3
"""Per-layer HBM traffic of one ResNet-50 pose forward against the bytes its tensors hold, from
the two counter passes of `tools/collect_traffic.sh infer` (run with PE_RESNET_DOWN_STREAM=0 so
launches are enqueued in layer order).

    python tools/infer_traffic_table.py [batch]

read bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies 64 B per 128-B request of a 16 B/lane
streaming read), written bytes = WRITE_SIZE x 1024.  "tensor bytes" = input + output
(+ residual) + weights, each once: what a layer moves if nothing is read twice and nothing is
served by the last-level cache.  Writes the sum over the forward into profiles/traffic.json.
"""
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from pose_estimation_amd import model as pm  # noqa: E402


def layer_bytes(batch, H=256, W=192):
    stem, blocks, deconvs, final = pm.resnet50_pose_specs()
    out = []
    h, w = H // 2, W // 2
    # stem + max pool are one launch: input image, pooled output, weights
    out.append(("stem+pool", batch * (h * w * 16 + (h // 2) * (w // 2) * stem.cout) * 2
                + stem.cout * 4 * 64 * 2))
    h, w = h // 2, w // 2
    have_c1 = False
    for i, b in enumerate(blocks):
        s = b.c2.stride
        ho, wo = h // s, w // s
        if not have_c1:
            out.append((b.c1.name, batch * h * w * (b.c1.cin + b.c1.cout) * 2 + b.c1.cin * b.c1.cout * 2))
        have_c1 = False
        if b.c2.cin == 64 and b.c2.cout == 64 and s == 1:
            # conv2 + conv3 (+ the projection of the skip path) are one launch: conv2's input,
            # the residual (or the projection's 64-channel input), the output, the weights;
            # behind a block without a projection the launch also holds the next block's conv1
            # (its 64-channel output and its weights on top)
            wts = (9 * 64 * 64 + 64 * b.c3.cout + (b.down.cin * b.down.cout if b.down else 0)) * 2
            res = b.down.cin if b.down else b.c3.cout
            label = b.c2.name.replace("conv2", "conv2+3") + ("+down" if b.down else "")
            nxt = blocks[i + 1].c1 if i + 1 < len(blocks) else None
            have_c1 = (b.down is None and nxt is not None and nxt.k == 1 and nxt.stride == 1
                       and nxt.cin == 256 and nxt.cout == 64)
            extra = 0
            if have_c1:
                label += "+next conv1"
                extra = batch * h * w * nxt.cout * 2 + nxt.cin * nxt.cout * 2
            out.append((label, batch * h * w * (64 + res + b.c3.cout) * 2 + wts + extra))
            continue
        out.append((b.c2.name, batch * (h * w * b.c2.cin + ho * wo * b.c2.cout) * 2
                    + 9 * b.c2.cin * b.c2.cout * 2))
        if b.down:
            out.append((b.down.name, batch * (h * w * b.down.cin + ho * wo * b.down.cout) * 2
                        + b.down.cin * b.down.cout * 2))
        out.append((b.c3.name, batch * ho * wo * (b.c3.cin + 2 * b.c3.cout) * 2
                    + b.c3.cin * b.c3.cout * 2))
        h, w = ho, wo
    for d in deconvs:
        out.append((d.name, batch * (h * w * d.cin + 4 * h * w * d.cout) * 2 + 16 * d.cin * d.cout * 2))
        h, w = 2 * h, 2 * w
    out.append(("final", batch * h * w * (final.cin * 2 + final.cout * 4) + final.cin * final.cout * 2))
    return out


def per_dispatch(path, counter):
    rows = [r for r in csv.DictReader(open(path))
            if any(k in r["Kernel_Name"] for k in ("igemm_kernel", "stem_pool_kernel",
                                                   "conv3x3_c64_kernel"))
            and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    d = ROOT / "gpurun_out" / "traffic" / "infer"
    fetch = per_dispatch(d / "fetch" / "p_counter_collection.csv", "FETCH_SIZE")
    write = per_dispatch(d / "write" / "p_counter_collection.csv", "WRITE_SIZE")
    layers = layer_bytes(batch)
    n = len(layers)
    assert len(fetch) % n == 0 and len(write) % n == 0, (len(fetch), len(write), n)
    fetch, write = fetch[-n:], write[-n:]
    tot_t = tot_m = 0
    print(f"{'layer':22s} {'read MB':>9s} {'written MB':>10s} {'tensor MB':>10s} {'ratio':>6s}")
    for (name, tb), f, w in zip(layers, fetch, write):
        rd, wr = 2 * f * 1024, w * 1024
        tot_t += tb
        tot_m += rd + wr
        print(f"{name:22s} {rd / 1e6:9.1f} {wr / 1e6:10.1f} {tb / 1e6:10.1f} {(rd + wr) / tb:6.2f}")
    print(f"{'forward':22s} {'':9s} {tot_m / 1e6:10.1f} {tot_t / 1e6:10.1f} {tot_m / tot_t:6.2f}")
    tfile = ROOT / "profiles" / "traffic.json"
    data = json.loads(tfile.read_text())
    data[f"resnet50_pose_infer_256x192_b{batch}_bf16"] = {
        "kernel": "igemm_kernel + stem_pool_kernel + conv3x3_c64_kernel (sum over the conv/deconv launches of one forward)",
        "command": "PE_RESNET_DOWN_STREAM=0 tools/collect_traffic.sh infer; tools/infer_traffic_table.py",
        "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950), written = WRITE_SIZE x 1024",
        "traffic_bytes_per_launch": int(tot_m),
        "algorithmic_bytes_per_launch": int(tot_t),
        "note": "per step; tensor bytes = every launch's input + output (+ residual) + weights once; "
                "stem + max pool, and conv2 + conv3 (+ skip projection) of layer1's bottlenecks, are "
                "one launch each (the tensors between them are never written)",
        "round": 3,
    }
    tfile.write_text(json.dumps(data, indent=1) + "\n")


if __name__ == "__main__":
    main()
