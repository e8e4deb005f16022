#!/bin/bash
# Round-3 evidence on the final sources: kernel statistics of the headline bench and of the other configs, PMC passes,
# the data-parallel step at world 1.  Everything lands in gpurun_out/evidence/ (copied into profiles/ afterwards).
set -o pipefail
mkdir -p gpurun_out/evidence; export TMPDIR=/tmp
E=$GRAFT_REPO_ROOT/gpurun_out/evidence
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
rm -rf gpurun_out/ev_prof1 gpurun_out/ev_prof2
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ev_prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > $E/r03_bench.json.log 2>&1); echo "rocprof headline exit $?"
cp $(find gpurun_out/ev_prof1 -name "*kernel_stats.csv" | head -1) $E/r03_bench_kernel_stats.csv
python3 tools/k1_split.py gpurun_out/ev_prof1 | tee $E/r03_bench_k1_split.txt
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ev_prof2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $E/r03_other_configs_bench.json.log 2>&1); echo "rocprof other configs exit $?"
cp $(find gpurun_out/ev_prof2 -name "*kernel_stats.csv" | head -1) $E/r03_other_configs_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE MfmaUtil MfmaFlopsBF16; do
  rm -rf gpurun_out/pmc_$c
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1)
  echo "== $c exit $?"
done
python3 tools/pmc_summarize.py r03 | tee $E/r03_pmc_summary.txt
cp profiles/r03_pmc_hbm_traffic.json profiles/r03_pmc_mfma.json $E/ 2>/dev/null
bash tools/dp_round.sh > $E/r03_dp_round.txt 2>&1; echo "dp exit $?"; cp gpurun_out/dp_seq.txt $E/r03_dp_world1_kernel_sequence.txt
cp gpurun_out/bench_dp_factors.log $E/r03_dp_world1_factors_bench.json.log; cp gpurun_out/bench_dp_allreduce.log $E/r03_dp_world1_allreduce_bench.json.log
tail -4 $E/r03_dp_round.txt
timeout -k 10 200 python tools/factor_probe.py 2>&1 | grep -v "^\[build\]" > $E/r03_dp_apply_factors_1_2_4_8_ranks.txt; echo "factor probe exit $?"; cat $E/r03_dp_apply_factors_1_2_4_8_ranks.txt
timeout -k 10 200 python tools/stamps_probe.py 2>&1 | grep -v "^\[build\]" > $E/r03_stamps.txt; echo "stamps exit $?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $E/r03_driver_form_bench.json.log 2>&1; echo "driver-form bench exit $?"
tail -c 600 $E/r03_driver_form_bench.json.log
