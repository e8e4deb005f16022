#!/bin/bash
# decode / represent of the image stack: wall time and kernel sequence
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
if [ -n "$1" ]; then timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "$1" > gpurun_out/pytest_dec.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_dec.log; fi
cd /tmp && timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/decode_probe.py 256 both | tee $GRAFT_REPO_ROOT/gpurun_out/decode_probe.txt
for w in decode represent; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_dec
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dec -- python3 $GRAFT_REPO_ROOT/tools/decode_probe.py 256 $w > $GRAFT_REPO_ROOT/gpurun_out/prof_dec.log 2>&1; echo "rocprof exit $?"
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_dec -name "*kernel_trace.csv" | head -1); python3 $GRAFT_REPO_ROOT/tools/trace_seq.py $f 12 | tee $GRAFT_REPO_ROOT/gpurun_out/decode_seq_$w.txt
done
