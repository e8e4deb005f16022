#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks of the last build (build/build.log):
one line per kernel with demangled name, VGPRs, AGPRs, SGPRs, scratch, LDS, occupancy."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
log = os.path.join(ROOT, "multimodal-idbn_amd", "build", "build.log")
txt = open(sys.argv[1] if len(sys.argv) > 1 else log).read()
keys = ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")
rows, cur = [], None
for ln in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for k in keys:
        m = re.search(re.escape(k) + r": (\d+)", ln)
        if m and k not in cur:
            cur[k] = int(m.group(1))
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
except Exception:
    dem = names
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS':>7} {'occ':>4}  kernel")
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*\)$", "", d).replace("imdbn::", "").replace("void ", "")
    print(f"{r.get(keys[0], 0):5d} {r.get(keys[1], 0):5d} {r.get(keys[2], 0):5d} {r.get(keys[3], 0):8d} {r.get(keys[5], 0):7d} {r.get(keys[4], 0):4d}  {d}")
