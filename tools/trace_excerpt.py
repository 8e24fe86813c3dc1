#!/usr/bin/env python3
"""Timeline excerpt of the last training step in a rocprofv3 kernel trace (CSV): per queue the
launch count and busy time, the launches shorter than 12 us on the busiest queue, and two
windows of consecutive launches (start, duration, queue, grid, kernel) that show dependent
launches starting the instant their predecessor ends.

    python tools/trace_excerpt.py <kernel_trace.csv> [first_index count] ...
"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    step = rows[adam[-2] + 1:adam[-1] + 1]
    t0 = int(step[0]["Start_Timestamp"])
    span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
    print(f"last step: {len(step)} launches, {span:.1f} us from first start to last end")
    byq = defaultdict(list)
    for r in step:
        byq[r["Queue_Id"]].append(r)
    for q, rs in sorted(byq.items()):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
        print(f"queue {q}: {len(rs)} launches, {busy:.1f} us busy")
    main_q = max(byq.values(), key=len)
    small = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in main_q]
    small = [d for d in small if d < 12.0]
    print(f"busiest queue: {len(small)} launches shorter than 12 us, {sum(small):.1f} us together")
    wins = sys.argv[2:] or ["0", "30", "200", "40"]
    for a, n in zip(wins[0::2], wins[1::2]):
        print(f"-- launches {a} .. {int(a) + int(n) - 1}: start us, duration us, queue, threads, kernel")
        for r in step[int(a):int(a) + int(n)]:
            st = (int(r["Start_Timestamp"]) - t0) / 1e3
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            name = r["Kernel_Name"].replace("void ", "").replace("pe::", "").split("(")[0]
            print(f"{st:9.1f} {d:7.1f}  q{r['Queue_Id']} {r['Grid_Size_X']:>8s}  {name[:70]}")


if __name__ == "__main__":
    main()
