#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
rm -rf gpurun_out/prof_205
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_205 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_205.log 2>&1); echo "rocprof exit $?"
tail -1 gpurun_out/prof_205.log | cut -c1-400
python3 tools/step_timeline.py gpurun_out/prof_205 20
