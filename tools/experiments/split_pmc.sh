#!/bin/bash
# counters-only pass over the split-precision strip forward: f16 matrix MOPs per launch against the
# flop model (utils/flop_model.py strip_split_mfma_issued); writes gpurun_out/split_pmc.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp SPLIT_ONLY=1; cd /tmp; rm -rf /tmp/rp_split
for ctr in SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "/tmp/rp_split/${ctr// /_}" -- python $ROOT/tools/experiments/split_strips_check.py > "/tmp/rp_split_${ctr// /_}.log" 2>&1 || tail -3 "/tmp/rp_split_${ctr// /_}.log"
  python $ROOT/tools/pmc_summary.py $(dirname $(find "/tmp/rp_split/${ctr// /_}" -name '*counter_collection.csv' | head -1)) 2>&1 | grep "strip_kernel"
done > $ROOT/gpurun_out/split_pmc.txt
cat $ROOT/gpurun_out/split_pmc.txt
