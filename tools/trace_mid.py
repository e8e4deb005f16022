#!/usr/bin/env python3
"""Kernel sequence around the middle occurrence of a named kernel in a rocprofv3 kernel trace (name, duration, gap)."""
import csv, re, sys, glob
f = sys.argv[1]; pat = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
i0 = max(0, idx[int(len(idx) * 0.8)] - 6)
prev = None
for r in rows[i0:i0 + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("imdbn::", "")[:60]
    print(f"{(s - prev) / 1e3 if prev else 0:8.2f} gap {(e - s) / 1e3:8.2f} us {nm}")
    prev = e
