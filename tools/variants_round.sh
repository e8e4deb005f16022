#!/bin/bash
# the GPU parity suite under engine-option variants (fallback paths), then a long soak of the bench loop
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for o in no_prefetch=1 no_bits=1 down_rows=32 no_chain_kernel=1 generic_k3=1 no_rank_loop=1 no_rank_acc=1; do
  IMDBN_OPTS="$o" timeout -k 10 400 python -m pytest tests -m gpu -q -x --timeout=300 > gpurun_out/pytest_$o.log 2>&1
  rc=$?; echo "== $o: exit $rc  $(tail -1 gpurun_out/pytest_$o.log)"
  if [ $rc -ge 124 ]; then echo "killed: stop"; exit $rc; fi
done
timeout -k 10 300 python bench.py --steps 20000 --warmup 50 --no-cpu-baseline | cut -c1-230
