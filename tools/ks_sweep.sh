for ks in 5 8 10; do
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ks$ks -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --no-prefetch --opt k1s_ks=$ks > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(ls -t $(find gpurun_out/prof_ks$ks -name "*kernel_stats.csv") | head -1)
  echo "== ks=$ks"; grep "k1_stream\|k2_stream\|assoc_update\|prep_operand" $f | awk -F, '{printf "%s calls %s avg %.2f us\n", substr($1,1,50), $2, $4/1000}'
done
