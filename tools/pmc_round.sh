#!/bin/bash
# Hardware counters for the bench kernels, one --pmc pass each, no trace domains mixed in:
# HBM traffic (FETCH_SIZE, WRITE_SIZE) and matrix-core use (MfmaUtil = busy cycles / SIMD cycles, MfmaFlopsBF16).
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE MfmaUtil MfmaFlopsBF16; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1)
  echo "== $c exit $?"; find gpurun_out/pmc_$c -name "*.csv" | head -5
done
python3 tools/pmc_summarize.py
