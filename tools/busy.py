#!/usr/bin/env python3
"""GPU busy fraction and top kernels of the last N ms of a rocprofv3 kernel trace."""
import csv, re, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
t_end = int(rows[-1]["End_Timestamp"]); t0 = t_end - int(win_ms * 1e6)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
busy = 0; last = t0; per = collections.Counter(); cnt = collections.Counter()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += max(0, e - max(s, last)); last = max(last, e)
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("imdbn::", "")[:60]
    per[nm] += e - s; cnt[nm] += 1
print(f"window {win_ms:.0f} ms: {len(rows)} kernels, GPU busy {100 * busy / (t_end - t0):.1f} %")
for nm, t in per.most_common(14):
    print(f"  {100 * t / (t_end - t0):5.1f} %  {cnt[nm]:6d} x {t / cnt[nm] / 1e3:8.2f} us  {nm}")
