#!/bin/bash
# kernel stats of the eager training step only (writes gpurun_out/prof_train/)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/prof_train; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/rp_train
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_train -o train -- python $ROOT/tools/train_step_profile.py eager_fused 20 > $OUT/train_run.log 2>&1
db=$(find /tmp/rp_train -name '*_results.db' | head -1)
[ -n "$db" ] && python $ROOT/tools/rocpd_kernel_stats.py $db $OUT/train_kernel_stats.csv
tail -2 $OUT/train_run.log
