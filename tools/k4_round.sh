#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 200 python tools/k4_probe.py "$@" > gpurun_out/k4_probe.log 2>&1; echo "probe exit $?"; grep -v amdgpu.ids gpurun_out/k4_probe.log | tail -12
