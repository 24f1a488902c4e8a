#!/bin/bash
# the round's profiles in one GPU call (writes gpurun_out/prof_r06/)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
bash tools/profile_round.sh r06 > gpurun_out/prof_r06_round.log 2>&1
python tools/pmc_forward_profile.py gpurun_out/prof_r06/pmc > gpurun_out/prof_r06_pmc.log 2>&1
OUT=$PWD/gpurun_out/prof_r06
export TMPDIR=/tmp
cd /tmp
for mode in compact sym; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/rp_k_${mode}_$ctr
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/rp_k_${mode}_$ctr -- \
      python /root/repo/tools/bench_lanczos_large.py --$mode --reps 2 > $OUT/kstep_${mode}_${ctr}_run.log 2>&1
    python /root/repo/tools/pmc_summary.py $(dirname $(find /tmp/rp_k_${mode}_$ctr -name '*counter_collection.csv' | head -1)) \
      > $OUT/kstep_${mode}_${ctr}_pmc.txt 2>&1
  done
done
cd /root/repo
for f in "--compact" "--sym" ""; do python tools/bench_lanczos_large.py $f 2>/dev/null | tail -1; done > $OUT/kstep_modes.jsonl
python tools/bench_train_step.py 2>/dev/null | tail -1 > $OUT/train_step.json
python tools/experiments/mid_check.py 2>/dev/null | tail -1 > $OUT/mid_check.json
ls $OUT | head -60; tail -3 gpurun_out/prof_r06_pmc.log
