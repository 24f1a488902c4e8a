cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
export TMPDIR=/tmp
for v in base pp; do (timeout 200 python tools/experiments/f32_linear_variants.py $v 2>&1 | tail -16) > gpurun_out/c4/f32_$v.log 2>&1; done
(timeout 300 python tools/experiments/f32_linear_stamps.py 2>&1 | tail -8) > gpurun_out/c4/stamps.log 2>&1
cat gpurun_out/c4/f32_*.log gpurun_out/c4/stamps.log
