#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for v in "$@"; do
  tag=$(echo "$v" | tr ' =-' '___')
  rm -rf gpurun_out/prof_exp
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_exp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs $v > $GRAFT_REPO_ROOT/gpurun_out/prof_exp_$tag.log 2>&1); echo "== $v : rocprof exit $?"
  f=$(ls -t $(find gpurun_out/prof_exp -name "*kernel_stats.csv") | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "imdbn" in n and int(r["Calls"]) > 5:
        print(f'   {n[:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f} max {float(r["MaxNs"])/1e3:7.2f}')
PY
  python3 tools/k1_split.py gpurun_out/prof_exp
done
