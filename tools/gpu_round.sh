#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel stats.  Outputs under gpurun_out/.
# usage: tools/gpu_round.sh [pytest -k expression]
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
if [ -n "$1" ]; then KEXPR=(-k "$1"); else KEXPR=(); fi
timeout -k 10 800 python -m pytest tests -m gpu -q -x --timeout=400 "${KEXPR[@]}" > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest exit $rc" | tee -a gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then tail -40 gpurun_out/pytest_gpu.log; echo "pytest was killed: no further GPU step in this call"; exit $rc; fi
tail -40 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "PYTEST FAILED: skipping bench"; exit $rc; fi
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -2 gpurun_out/bench.log | cut -c1-1500
# kernel trace of the same bench command (summary copied to profiles/ by hand)
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-180
