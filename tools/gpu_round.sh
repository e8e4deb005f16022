#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel stats.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 500 python -m pytest tests -m gpu -q -x --timeout=300 > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest exit $rc" | tee -a gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then tail -40 gpurun_out/pytest_gpu.log; echo "pytest was killed: no further GPU step in this call"; exit $rc; fi
tail -40 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "PYTEST FAILED: skipping bench"; exit $rc; fi
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
timeout -k 10 200 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -2 gpurun_out/bench.log
# kernel trace of the same bench command (summary copied to profiles/ by hand)
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-180
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --force-dp > gpurun_out/bench_dp1.log 2>&1; echo "bench force-dp exit $?"; tail -1 gpurun_out/bench_dp1.log | cut -c1-260
