#!/usr/bin/env python3
"""Print the steady-state kernel sequence of a rocprofv3 kernel trace: name, duration, gap to the previous kernel's end."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tail = rows[-n:]
prev = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("imdbn::", "")[:70]
    print(f"{(s - prev) / 1e3 if prev else 0:8.2f} gap  {(e - s) / 1e3:8.2f} us  grid {r.get('Grid_Size','?'):>8} wg {r.get('Workgroup_Size','?'):>5}  {nm}")
    prev = e
