#!/usr/bin/env python3
"""Per-layer time and TFLOP/s of one inference forward, from a rocprofv3 kernel trace of
`bench.py --workload infer` (CSV with --kernel-trace --output-format csv).

    python tools/infer_layer_table.py <kernel_trace.csv> [batch]

The last forward in the trace is tabulated; a transposed conv is one launch (four phases).
"""
import csv
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from pose_estimation_amd import flops  # noqa: E402
from pose_estimation_amd import model as pm  # noqa: E402


def main():
    path = sys.argv[1]
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    tr = list(csv.DictReader(open(path)))
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the launches that carry a layer: the engine, the stem + pool kernel (one launch for the
    # stem) and the resident-filter 3x3 kernel
    ig = [r for r in tr if any(k in r["Kernel_Name"] for k in
                               ("igemm", "stem_pool_kernel", "conv3x3_c64_kernel"))]
    _, rows, _ = flops.forward_macs()
    macs = dict(rows)
    stem, blocks, deconvs, final = pm.resnet50_pose_specs()
    # a launch may cover several layers: conv2 + conv3 (+ the projection of the skip path) of
    # the bottlenecks whose 3x3 layer has 64 channels go through one fused launch
    # ... and, behind a block without a projection, the next block's conv1 (256 -> 64) as well
    seq = [("stem", ["stem"])]
    have_c1 = False
    for i, b in enumerate(blocks):
        if b.c2.cin == 64 and b.c2.cout == 64 and b.c2.stride == 1:
            group = [b.c2.name, b.c3.name] + ([b.down.name] if b.down else [])
            label = b.c2.name.replace("conv2", "conv2+3") + ("+down" if b.down else "")
            if not have_c1:
                seq.append((b.c1.name, [b.c1.name]))
            nxt = blocks[i + 1].c1 if i + 1 < len(blocks) else None
            have_c1 = (b.down is None and nxt is not None and nxt.k == 1 and nxt.stride == 1
                       and nxt.cin == 256 and nxt.cout == 64)
            if have_c1:
                group.append(nxt.name)
                label += "+next conv1"
            seq.append((label, group))
        else:
            seq += [(n, [n]) for n in [b.c1.name, b.c2.name] + ([b.down.name] if b.down else [])
                    + [b.c3.name]]
    seq += [(d.name, [d.name]) for d in deconvs] + [("final", ["final"])]
    assert len(ig) % len(seq) == 0, (len(ig), len(seq))
    last = ig[-len(seq):]
    tot = 0.0
    for (name, group), r in zip(seq, last):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += dur
        tf = 2 * sum(macs[g] for g in group) * batch / (dur * 1e-6) / 1e12
        wgs = int(r["Grid_Size_X"]) // 256 * int(r["Grid_Size_Y"])
        kn = r["Kernel_Name"]
        kern = kn.split("<")[1].split(">")[0] if "<" in kn else kn.split("(")[0].replace("pe::", "")
        print(f"{name:34s} tile {kern:8s} wgs {wgs:6d} {dur:8.1f} us {tf:7.1f} TF/s")
    print(f"layer kernels total {tot:.1f} us (the stem's row includes the max pool when the two are "
          "one launch)")


if __name__ == "__main__":
    main()
