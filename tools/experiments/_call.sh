# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04b
mkdir -p $OUT
cd /tmp
for name in bench train; do
  rm -rf /tmp/rp_$name
  if [ $name = bench ]; then cmd="python $GRAFT_REPO_ROOT/bench.py"; else cmd="python $GRAFT_REPO_ROOT/tools/train_step_profile.py eager_fused 20"; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o $name -- $cmd > $OUT/${name}_run.log 2>&1
  db=$(find /tmp/rp_$name -name '*_results.db' | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_kernel_stats.py $db $OUT/${name}_kernel_stats.csv > /dev/null
done
grep '^{' $OUT/bench_run.log | tail -1 > $OUT/bench.json
tail -2 $OUT/train_run.log
ls -la $OUT
