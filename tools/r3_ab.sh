#!/bin/bash
# A/B of bench variants in ONE box (devices differ by a few %): untagged vs tagged batches, 20/5 vs 200/20
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs "$@" > gpurun_out/ab_$name.log 2>&1; 
  tail -1 gpurun_out/ab_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'updates/s', round(d['value'],1), 'us/step', round(1e3*d['ms_per_step'],2), 'enqueue us', round(d['host_enqueue_us_per_step'],1), d['host_enqueue_us_p50_max'], 'K3 us', d['roofline'] and round(d['roofline']['avg_launch_us'],2))"; }
for rep in 1 2; do
run untagged_20_5_$rep --steps 20 --warmup 5
run untagged_200_20_$rep --steps 200 --warmup 20
run tagged_200_20_$rep --steps 200 --warmup 20 --tag-batches
run tagged_20_5_$rep --steps 20 --warmup 5 --tag-batches
done
