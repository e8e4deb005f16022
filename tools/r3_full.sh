#!/bin/bash
# Full GPU suite, smoke, then the driver-form bench (defaults: cpu baseline + other configs).
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout=400 > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest exit $rc" | tee -a gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "PYTEST FAILED: skipping bench"; exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_form.log 2>&1; echo "bench exit $?"
tail -1 gpurun_out/bench_driver_form.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('updates/s', round(d['value'],1), 'us/step', round(1e3*d['ms_per_step'],2), 'roofline', d['roofline'])
print('cpu', d['cpu_baseline'])
for k,v in d.get('other_configs',{}).items(): print(k, json.dumps(v))
"
