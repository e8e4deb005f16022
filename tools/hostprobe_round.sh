#!/bin/bash
# Is the occasional host-bound first bench run an artefact of thread oversubscription / cold start?
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import torch; print('torch threads', torch.get_num_threads())"
for i in 1 2 3; do
  timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('run', d['value'], d['host_enqueue_us_per_step'], d['host_enqueue_us_p50_max'])" || exit 1
done
for i in 1 2; do
  OMP_NUM_THREADS=4 timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('omp4', d['value'], d['host_enqueue_us_per_step'], d['host_enqueue_us_p50_max'])" || exit 1
done
timeout -k 10 120 python bench.py --steps 3000 --warmup 30 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('long', d['value'], d['host_enqueue_us_per_step'], d['host_enqueue_us_p50_max'])"
