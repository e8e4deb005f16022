#!/bin/bash
# Kernel trace of the bench command + per-block stamps of K1 / K2.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
rm -rf gpurun_out/prof
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
f=$(ls -t $(find gpurun_out/prof -name "*kernel_stats.csv") | head -1)
python3 - "$f" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "imdbn" in n:
        print(f'{n[:100]:100s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f} max {float(r["MaxNs"])/1e3:7.2f}')
PY
timeout -k 10 200 python tools/stamps_probe.py 2>&1 | grep -v "^\[build\]" | head -24
