#!/bin/bash
# A/B one engine option: parity tests under IMDBN_OPTS=$1, then kernel stats with and without it.
#   bash tools/ab_round.sh k1x=1
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
IMDBN_OPTS="$1" timeout -k 10 500 python -m pytest tests -m gpu -q -x --timeout=300 > gpurun_out/pytest_ab.log 2>&1
rc=$?; tail -15 gpurun_out/pytest_ab.log
if [ $rc -ne 0 ]; then echo "PYTEST FAILED under $1"; exit $rc; fi
bash tools/dbg_round.sh dbg=0 "$1"
for o in dbg=0 "$1"; do
  args=""; for kv in $(echo $o | tr ',' ' '); do args="$args --opt $kv"; done
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline $args | cut -c1-200
done
