cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c16
export TMPDIR=/tmp
for v in base wn2 base wn2; do (timeout 120 python tools/experiments/f32_linear_variants.py $v 2>&1 | grep 'hand\|CHECK' | cut -c1-190) >> gpurun_out/c16/f32.log 2>&1; done
cat gpurun_out/c16/f32.log
