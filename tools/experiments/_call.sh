cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
export TMPDIR=/tmp
for v in base base; do (timeout 120 python tools/experiments/f32_linear_variants.py $v 2>&1 | grep 'hand\|check\|CHECK') >> gpurun_out/c7/f32.log 2>&1; done
(timeout 600 python -m pytest tests/test_gpu_ada.py -m gpu -x -q -k "f32_linear or folded_filter or config4 or end_to_end" 2>&1 | tail -4) > gpurun_out/c7/pytest.log 2>&1
cat gpurun_out/c7/f32.log | cut -c1-210; tail -3 gpurun_out/c7/pytest.log
