#!/usr/bin/env python3
"""Averages rocprofv3 counter_collection.csv files per (kernel, counter).
    python tools/pmc_table.py <dir with p*/p_counter_collection.csv> [kernel substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "igemm"
acc = defaultdict(list)
for f in sorted(glob.glob(f"{d}/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:36s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
