#!/usr/bin/env python3
"""Idle time between consecutive kernels of each HIP queue in a rocprofv3 kernel trace
(--kernel-trace --output-format csv), for the last full step of a bench run.

    python tools/stream_gaps.py <kernel_trace.csv> [marker-kernel-substring]

The step is delimited by two launches of the marker kernel (default: adam_kernel)."""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    a, b = marks[-2], marks[-1]
    step = rows[a + 1:b + 1]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    print(f"step: {len(step)} launches, {(t1 - t0) / 1e6:.3f} ms from first start to last end")
    by_q = defaultdict(list)
    for r in step:
        by_q[r["Queue_Id"]].append(r)
    for q, rs in by_q.items():
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
        gaps = [int(n["Start_Timestamp"]) - int(p["End_Timestamp"]) for p, n in zip(rs, rs[1:])]
        pos = [g for g in gaps if g > 0]
        span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])
        print(f"queue {q}: {len(rs)} launches, busy {busy / 1e6:.3f} ms over a span of "
              f"{span / 1e6:.3f} ms; {len(pos)} gaps, {sum(pos) / 1e6:.3f} ms idle, "
              f"median gap {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.2f} us")


if __name__ == "__main__":
    main()
