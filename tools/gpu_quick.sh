#!/bin/bash
# Quick GPU iteration: a subset of the parity tests, then the bench line (no profile).  usage: tools/gpu_quick.sh ["pytest -k expr"] [bench args]
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
K="${1:-headline or philox_cd_step or c2_headline or prefetch or random_shapes or full_size}"
timeout -k 10 500 python -m pytest tests -m gpu -q -x --timeout=300 -k "$K" > gpurun_out/pytest_quick.log 2>&1
rc=$?; tail -15 gpurun_out/pytest_quick.log; echo "pytest exit $rc"
if [ $rc -ne 0 ]; then exit $rc; fi
shift
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs "$@" > gpurun_out/bench_quick.log 2>&1; echo "bench exit $?"
tail -1 gpurun_out/bench_quick.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('updates/s', round(d['value'],1), 'us/step', round(1e3*d['ms_per_step'],2), 'K3 us', d['roofline'] and round(d['roofline']['avg_launch_us'],2))"
