#!/bin/bash
# headline step time and K3 launch time for several absolute row pitches of W / W_m (floats)
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for p in "$@"; do
IMDBN_ROW_PITCH_ABS=$p timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > gpurun_out/bench_pitch_$p.log 2>&1 || { echo "pitch $p failed"; tail -3 gpurun_out/bench_pitch_$p.log; continue; }
tail -1 gpurun_out/bench_pitch_$p.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pitch $p: us/step', round(1e3*d['ms_per_step'],2), 'K3 us', round(d['roofline']['avg_launch_us'],2))"
done
