#!/bin/bash
# kernel stats of bench.py under different engine options (imdbn_set_option name=value, comma separated), e.g.
#   bash tools/dbg_round.sh down_rows=32 no_bits=1 ksplit_up=10
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for o in "$@"; do
  rm -rf gpurun_out/dbg_$(echo "$o" | tr "= ," "___")
  tag=$(echo "$o" | tr '= ,' '___')
  args=""; for kv in $(echo $o | tr ',' ' '); do args="$args --opt $kv"; done
  (cd /tmp && timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dbg_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline $args > $GRAFT_REPO_ROOT/gpurun_out/dbg_$tag.log 2>&1)
  echo "== $o"; rm -f /tmp/ks.csv; find gpurun_out/dbg_$tag -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} /tmp/ks.csv
  python3 - <<'PY'
import csv, os
if os.path.exists('/tmp/ks.csv'):
    for r in list(csv.DictReader(open('/tmp/ks.csv')))[:8]:
        print(f"{r['Name'][:64]:64s} {r['Calls']:>5s} {float(r['AverageNs'])/1000:8.1f} us")
PY
done
