#!/usr/bin/env python3
"""Markdown table of the counter passes of tools/pmc_mfma.sh.
    python tools/mfma_pmc_table.py <dir of <shape>.csv counter collections>"""
import csv
import glob
import os
import sys
from collections import defaultdict

print("| shape B,H,W,Cin,Cout,k | launches | SQ_INSTS_MFMA | MFMA busy cycles | kernel cycles | "
      "MFMA utilisation | waiting share of wave cycles | LDS bank conflict cycles |")
print("|---|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.csv"))):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "igemm" in r["Kernel_Name"] or "conv3x3_c64" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    n = len(acc["SQ_INSTS_MFMA"])
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
    wait = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    shape = os.path.basename(f)[:-4].replace("_", ",")
    print(f"| {shape} | {n} | {m['SQ_INSTS_MFMA']:.0f} | {m['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} | "
          f"{cyc:.0f} | {100 * util:.1f} % | {100 * wait:.1f} % | {m['SQ_LDS_BANK_CONFLICT']:.0f} |")
