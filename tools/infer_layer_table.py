#!/usr/bin/env python3
"""Per-layer time and TFLOP/s of one inference forward, from a rocprofv3 kernel trace of
`bench.py --workload infer` (CSV with --kernel-trace --output-format csv).

    python tools/infer_layer_table.py <kernel_trace.csv> [batch] [steps_in_trace]
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
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    tr = list(csv.DictReader(open(path)))
    ig = [r for r in tr if "igemm" in r["Kernel_Name"]]
    per_fwd = len(ig) // steps
    last = ig[-per_fwd:]
    _, rows, _ = flops.forward_macs()
    macs = {}
    for n, m in rows:
        if n.startswith("deconv"):
            for i in range(4):
                macs[f"{n}.p{i}"] = m / 4
        else:
            macs[n] = m
    stem, blocks, deconvs, final = pm.resnet50_pose_specs()
    seq = ["stem"]
    for b in blocks:
        seq += [b.c1.name, b.c2.name] + ([b.down.name] if b.down else []) + [b.c3.name]
    for d in deconvs:
        seq += [f"{d.name}.p{i}" for i in range(4)]
    seq.append("final")
    assert len(seq) == per_fwd, (len(seq), per_fwd)
    tot = 0.0
    for name, r in zip(seq, last):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += dur
        tf = 2 * macs[name] * batch / (dur * 1e-6) / 1e12
        print(f"{name:18s} wgs {int(r['Grid_Size_X']) // 256:6d} {dur:8.1f} us {tf:7.1f} TF/s")
    print(f"igemm total {tot:.1f} us")


if __name__ == "__main__":
    main()
