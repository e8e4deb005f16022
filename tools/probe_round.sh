#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for m in "$@"; do
  (cd /tmp && timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/probe_$m -- python3 $GRAFT_REPO_ROOT/tools/finish_probe.py $m > $GRAFT_REPO_ROOT/gpurun_out/probe_$m.log 2>&1)
  echo "== $m"; f=$(find gpurun_out/probe_$m -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:7]:
    if 'imdbn' in r[0]: print(f"  {r[0][:58]:58s} {float(r[3])/1000:8.1f} us  x{r[1]}")
PY
done
