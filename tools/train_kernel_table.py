#!/usr/bin/env python3
"""Per-launch table of the last training step in a rocprofv3 kernel trace of
`bench.py --workload train` (CSV, --kernel-trace --output-format csv).

    python tools/train_kernel_table.py <kernel_trace.csv> [top]

Prints the launches of the last step in order (index, kernel, grid, duration) for the `top`
longest ones, and totals per kernel family.
"""
import csv
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "").replace("pe::", "")
    return name.split("(")[0]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step ends with the batched weight transpose; take what lies between the last two
    ends = [i for i, r in enumerate(rows) if "transpose_w_batched" in r["Kernel_Name"]]
    step = rows[ends[-2] + 1: ends[-1] + 1]
    t0 = int(step[0]["Start_Timestamp"])
    span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
    recs = []
    fam = defaultdict(float)
    for i, r in enumerate(step):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        g = f'{int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))}x{r["Grid_Size_Y"]}'
        recs.append((d, i, short(r["Kernel_Name"]), g))
        fam[short(r["Kernel_Name"])] += d
    busy = sum(d for d, *_ in recs)
    print(f"{len(step)} launches, span {span:.1f} us, busy {busy:.1f} us")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f"  {v:9.1f} us  {k}")
    print("longest launches:")
    for d, i, k, g in sorted(recs, reverse=True)[:top]:
        print(f"  #{i:4d} {d:8.1f} us  wgs {g:10s} {k}")


if __name__ == "__main__":
    main()
