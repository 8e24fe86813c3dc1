#!/usr/bin/env python3
"""HBM traffic of ONE step of a network workload, summed over every kernel of the step, from the
two counter passes of `tools/collect_traffic.sh <workload>`; written to profiles/traffic.json.

    python tools/step_traffic.py <workload> <kernel that ends a step, e.g. adam_kernel> [round]

A step is what lies between two consecutive dispatches of the step-ending kernel (the last two
of the run).  read bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies 64 B per 128-B request of a
16 B/lane streaming read), written bytes = WRITE_SIZE x 1024.
"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def one_step(path, counter, end_kernel):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ends = [i for i, r in enumerate(rows) if end_kernel in r["Kernel_Name"]]
    step = rows[ends[-2] + 1: ends[-1] + 1]
    fam = defaultdict(float)
    for r in step:
        name = r["Kernel_Name"].replace("void ", "").replace("pe::", "").split("(")[0].split("<")[0]
        fam[name] += float(r["Counter_Value"])
    return sum(fam.values()), fam, len(step)


def main():
    wl, end_kernel = sys.argv[1], sys.argv[2]
    rnd = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    d = ROOT / "gpurun_out" / "traffic" / wl
    f, ffam, nf = one_step(d / "fetch" / "p_counter_collection.csv", "FETCH_SIZE", end_kernel)
    w, wfam, nw = one_step(d / "write" / "p_counter_collection.csv", "WRITE_SIZE", end_kernel)
    name = json.loads([l for l in (d / "fetch" / "run.log").read_text().splitlines()
                       if l.startswith("{")][-1])["config"]["workload"]
    total = int(2 * f * 1024 + w * 1024)
    print(f"{name}: {nf} dispatches per step, read {2 * f * 1024 / 1e9:.3f} GB, "
          f"written {w * 1024 / 1e9:.3f} GB, total {total / 1e9:.3f} GB")
    for k in sorted(set(ffam) | set(wfam), key=lambda k: -(2 * ffam.get(k, 0) + wfam.get(k, 0)))[:14]:
        print(f"  {k[:44]:44s} read {2 * ffam.get(k, 0) * 1024 / 1e6:9.1f} MB   written "
              f"{wfam.get(k, 0) * 1024 / 1e6:9.1f} MB")
    # the step's algorithmic bytes (every pass reads its operands and writes its result once)
    sys.path.insert(0, str(ROOT))
    alg, note = None, "per step, all kernels"
    if name.startswith("resnet50_pose_train"):
        from pose_estimation_amd import flops
        alg = flops.train_step_bytes(64)[0]
        note += "; algorithmic bytes = pose_estimation_amd.flops.train_step_bytes(64)"
    elif name.startswith("hrnet_w32_pose_train"):
        from pose_estimation_amd import hrnet
        alg = hrnet.train_step_bytes(hrnet.hrnet_specs(), 64)[0]
        note += "; algorithmic bytes = pose_estimation_amd.hrnet.train_step_bytes(spec, 64)"
    if alg:
        print(f"  algorithmic {alg / 1e9:.3f} GB -> traffic / algorithmic = {total / alg:.2f}")
    tfile = ROOT / "profiles" / "traffic.json"
    data = json.loads(tfile.read_text())
    data[name] = {
        "kernel": f"every kernel of one step (ends with {end_kernel})",
        "command": f"tools/collect_traffic.sh {wl}; tools/step_traffic.py {wl} {end_kernel}",
        "dispatches": [nf, nw],
        "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950), written = WRITE_SIZE x 1024",
        "traffic_bytes_per_launch": total,
        "read_bytes": int(2 * f * 1024), "written_bytes": int(w * 1024),
        "algorithmic_bytes_per_launch": alg,
        "note": note,
        "round": rnd,
    }
    tfile.write_text(json.dumps(data, indent=1) + "\n")


if __name__ == "__main__":
    main()
