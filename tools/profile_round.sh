#!/bin/bash
# The profiles of a round in one GPU call (run through gpurun; writes gpurun_out/prof_<tag>/):
#   tools/profile_round.sh r04
# 1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (bench line + kernel stats)
# 2. the same of the LanczosNet training step (eager, fused Adam)
# 3. counters-only passes (--pmc with --kernel-trace, nothing else): matrix-pipe busy + clock of
#    lnz_f32_linear and the library GEMM on the filter-MLP shapes; FETCH_SIZE / WRITE_SIZE of the
#    large-graph conv kernels (folded and every-channel) — separate passes per counter group
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
stats() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o $name -- "$@" > $OUT/${name}_run.log 2>&1
  local db=$(find /tmp/rp_$name -name '*_results.db' | head -1)
  [ -n "$db" ] && python $ROOT/tools/rocpd_kernel_stats.py $db $OUT/${name}_kernel_stats.csv
}
pmc() {  # name, counters (quoted), command...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/rp_$name
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/rp_$name -- "$@" > $OUT/${name}_run.log 2>&1
  python $ROOT/tools/pmc_summary.py $(dirname $(find /tmp/rp_$name -name '*counter_collection.csv' | head -1)) > $OUT/${name}_pmc.txt 2>&1
}
stats bench python $ROOT/bench.py
grep '^{' $OUT/bench_run.log | tail -1 > $OUT/bench.json
stats train python $ROOT/tools/train_step_profile.py eager_fused 20
stats config5 python $ROOT/tools/bench_config5.py --reps 2
pmc f32lin "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" python $ROOT/tools/bench_f32_linear.py
pmc c5fetch "FETCH_SIZE" python $ROOT/tools/bench_config5.py --reps 1
pmc c5write "WRITE_SIZE" python $ROOT/tools/bench_config5.py --reps 1
ls -la $OUT
