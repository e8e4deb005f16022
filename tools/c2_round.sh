#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 200 python tools/c2_trace.py 4 > gpurun_out/c2_time.log 2>&1; tail -1 gpurun_out/c2_time.log
rm -rf gpurun_out/c2prof
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c2prof -- python3 $GRAFT_REPO_ROOT/tools/c2_trace.py 1 > $GRAFT_REPO_ROOT/gpurun_out/c2prof.log 2>&1; echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/c2prof -name "*kernel_trace.csv" | head -1); python tools/trace_seq.py $f 60 > gpurun_out/c2_seq.txt; tail -22 gpurun_out/c2_seq.txt
