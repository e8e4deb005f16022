#!/bin/bash
# kernel trace of the bench command, untagged vs tagged batches, same box
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for v in untagged tagged; do
  rm -rf gpurun_out/prof_$v
  extra=""; [ $v = tagged ] && extra="--tag-batches"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs $extra > $GRAFT_REPO_ROOT/gpurun_out/prof_$v.log 2>&1); echo "rocprof $v exit $?"
  f=$(ls -t $(find gpurun_out/prof_$v -name "*kernel_stats.csv") | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "imdbn" in n:
        print(f'   {n[:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f} max {float(r["MaxNs"])/1e3:7.2f}')
PY
done
