#!/usr/bin/env python3
"""gpurun_out/pmc_{FETCH_SIZE,WRITE_SIZE}/**/counter_collection.csv -> profiles/<round>_pmc_hbm_traffic.json (round tag = argv[1], default r03)
(HBM bytes per launch per kernel; gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request)."""
import collections, csv, glob, hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r03"
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = sorted(glob.glob(os.path.join(ROOT, f"gpurun_out/pmc_{c}/**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        sys.exit(f"no csv for {c}")
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if r.get("Counter_Name") == c and "imdbn" in r["Kernel_Name"]:
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][f"{c}_KB_mean"] = sum(v) / len(v)
        res[k]["launches"] = len(v)
for k, d in res.items():
    d["hbm_bytes_per_launch"] = (2 * d.get("FETCH_SIZE_KB_mean", 0.0) + d.get("WRITE_SIZE_KB_mean", 0.0)) * 1024
def _engine_source_sha():      # the same stamp bench.py computes: says which kernel sources the counters were taken on
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "multimodal-idbn_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


out = {
    "engine_source_sha": _engine_source_sha(),
    "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} (separate passes) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs",
    "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  [gfx950: FETCH_SIZE counts 64 B per 128-B request]",
    "history": "round 1 (profiles/r01_pmc_hbm_traffic.json): K1 gemm_up4_partial 87.2 MB + finish 13.5 MB, K2 gemm_down_fused_next 87.1 MB (16.4 MB of writes), K3 252.5 MB per launch; "
               "round 2 (profiles/r02_pmc_hbm_traffic.json): k1_stream 68.3 MB, k2_stream 66.5 MB, K3 252.7 MB",
    "kernels": dict(res),
}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{ROUND}_pmc_hbm_traffic.json"), "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
    print(f"{k[:60]:60s} {d['hbm_bytes_per_launch']/1e6:9.1f} MB/launch  x{d['launches']}")

# matrix-core counters (separate passes): MfmaUtil [% of SIMD cycles the MFMA pipe is busy], MfmaFlopsBF16 per launch
mf = collections.defaultdict(dict)
for c in ("MfmaUtil", "MfmaFlopsBF16"):
    fs = sorted(glob.glob(os.path.join(ROOT, f"gpurun_out/pmc_{c}/**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if r.get("Counter_Name") == c and "imdbn" in r["Kernel_Name"]:
            acc[re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        mf[k][c + "_mean"] = sum(v) / len(v)
        mf[k]["launches"] = len(v)
if mf:
    json.dump({"command": "rocprofv3 --pmc {MfmaUtil|MfmaFlopsBF16} (separate passes) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs",
               "note": "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs) * 100; the update is bandwidth bound, the MFMA pipe is mostly idle by design "
                       "(bf16 MFMA roofline of the whole update: 3.8 us of ~131 us)",
               "kernels": dict(mf)}, open(os.path.join(ROOT, "profiles", f"{ROUND}_pmc_mfma.json"), "w"), indent=1)
    for k, d in mf.items():
        print(f"{k[:60]:60s} MfmaUtil {d.get('MfmaUtil_mean', float('nan')):6.2f} %   bf16 flops/launch {d.get('MfmaFlopsBF16_mean', 0)/1e9:8.2f} G")
