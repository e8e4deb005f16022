#!/bin/bash
# Round-3 GPU round: [pytest -k expr] then the driver's bench command (20 / 5) and the long one (200 / 20).  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
if [ -n "$1" ]; then KEXPR=(-k "$1"); else KEXPR=(); fi
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout=400 "${KEXPR[@]}" > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest exit $rc" | tee -a gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "PYTEST FAILED: skipping bench"; exit $rc; fi
shift
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs "$@" > gpurun_out/bench_20_5.log 2>&1; echo "bench 20/5 exit $?"
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs "$@" > gpurun_out/bench_200_20.log 2>&1; echo "bench 200/20 exit $?"
for f in gpurun_out/bench_20_5.log gpurun_out/bench_200_20.log; do
tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', 'updates/s', round(d['value'],1), 'us/step', round(1e3*d['ms_per_step'],2), 'enqueue us', round(d['host_enqueue_us_per_step'],1), d['host_enqueue_us_p50_max'], 'K3 us', d['roofline'] and round(d['roofline']['avg_launch_us'],2))"
done
