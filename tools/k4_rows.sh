#!/bin/bash
# C3 / C5 timings (bench.py other_configs) for several rows-per-block settings of the chain kernel
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for r in 0 1 2 3; do
timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-k3-events --opt chain_rows=$r > gpurun_out/bench_k4_$r.log 2>&1 || { echo "rows $r failed"; tail -5 gpurun_out/bench_k4_$r.log; exit 1; }
tail -1 gpurun_out/bench_k4_$r.log | python -c "
import sys,json; d=json.loads(sys.stdin.read())['other_configs']
print('rows $r:', 'C2 %.1f us' % (1e3*d['C2_stack']['ms_per_batch']), 'C3 main %.3f warm %.3f ms' % (d['C3_joint_532x256']['main_step_ms'], d['C3_joint_532x256']['warmup_step_ms']), 'C5 k5 %.3f k16 %.3f decode %.3f ms' % (d['C5_cross_reconstruct_b256_s50']['ms_default_k5_inert'], d['C5_cross_reconstruct_b256_s50']['ms_live_k16'], d['C5_cross_reconstruct_b256_s50']['decode_ms']))"
done
