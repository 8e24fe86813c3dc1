#!/usr/bin/env python3
"""Folds the counter passes of tools/collect_traffic.sh into profiles/traffic.json.

    python tools/update_traffic.py <workload>=<kernel substring>=<algorithmic bytes> ...

Read bytes = 2 x FETCH_SIZE x 1024 (on gfx950 FETCH_SIZE tallies 64 B per 128-B request of a
16 B/lane streaming read), written bytes = WRITE_SIZE x 1024; means over the kernel's dispatches.
"""
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def mean_counter(path, sub, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


def main():
    tfile = ROOT / "profiles" / "traffic.json"
    data = json.loads(tfile.read_text())
    for arg in sys.argv[1:]:
        wl, sub, alg = arg.split("=")
        d = ROOT / "gpurun_out" / "traffic" / wl
        f, nf = mean_counter(d / "fetch" / "p_counter_collection.csv", sub, "FETCH_SIZE")
        w, nw = mean_counter(d / "write" / "p_counter_collection.csv", sub, "WRITE_SIZE")
        name = json.loads([l for l in (d / "fetch" / "run.log").read_text().splitlines()
                           if l.startswith("{")][-1])["config"]["workload"]
        data[name] = {
            "kernel": sub,
            "command": f"tools/collect_traffic.sh {wl}",
            "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "dispatches": [nf, nw],
            "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950), written = WRITE_SIZE x 1024",
            "traffic_bytes_per_launch": int(2 * f * 1024 + w * 1024),
            "algorithmic_bytes_per_launch": int(alg), "round": 1,
        }
        print(name, data[name]["traffic_bytes_per_launch"], "vs algorithmic", alg)
    tfile.write_text(json.dumps(data, indent=1) + "\n")


if __name__ == "__main__":
    main()
