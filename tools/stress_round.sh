#!/bin/bash
# randomised parity stress on the GPU box: default shapes, then the large-layer regime
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout -k 10 ${3:-500} python tools/stress_parity.py ${1:-120} ${2:-7} > gpurun_out/stress_small.log 2>&1; echo "stress small exit $?"; grep -v amdgpu.ids gpurun_out/stress_small.log | tail -12
STRESS_BIG=1 timeout -k 10 ${3:-500} python tools/stress_parity.py $(( ${1:-120} / 3 )) ${2:-7} > gpurun_out/stress_big.log 2>&1; echo "stress big exit $?"; grep -v amdgpu.ids gpurun_out/stress_big.log | tail -12
