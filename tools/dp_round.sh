#!/bin/bash
# The data-parallel step on one GPU (world 1 over RCCL): bench line + kernel sequence with gaps.
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for m in factors allreduce; do
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --force-dp --dp-mode $m > gpurun_out/bench_dp_$m.log 2>&1 || { echo "bench $m failed"; tail -5 gpurun_out/bench_dp_$m.log; exit 1; }
tail -1 gpurun_out/bench_dp_$m.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m: updates/s', round(d['value'],1), 'us/step', round(1e3*d['ms_per_step'],2), 'host enqueue us', round(d['host_enqueue_us_per_step'],1))"
done
rm -rf gpurun_out/dpprof
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dpprof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs --force-dp --no-k3-events > $GRAFT_REPO_ROOT/gpurun_out/dpprof.log 2>&1; echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/dpprof -name "*kernel_trace.csv" | head -1); python tools/trace_seq.py $f 36 > gpurun_out/dp_seq.txt; tail -30 gpurun_out/dp_seq.txt
