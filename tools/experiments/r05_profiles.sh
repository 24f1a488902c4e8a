#!/bin/bash
# the round's profiles in one GPU call (writes gpurun_out/prof_r05/)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
bash tools/profile_round.sh r05 > gpurun_out/prof_r05_round.log 2>&1
python tools/pmc_forward_profile.py gpurun_out/prof_r05/pmc > gpurun_out/prof_r05_pmc.log 2>&1
python tools/bench_ritz_sweep.py > gpurun_out/prof_r05/ritz32_batch_sweep.json 2> gpurun_out/prof_r05/ritz32_batch_sweep.err
python tools/bench_ritz_wg.py > gpurun_out/prof_r05/ritz_wg.jsonl 2>/dev/null
ls gpurun_out/prof_r05 | head -40; tail -3 gpurun_out/prof_r05_pmc.log
