cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/c1/pytest.log 2>&1
(timeout 300 python tools/bench_config5.py --reps 3 2>&1 | tail -5) > gpurun_out/c1/config5.log 2>&1
for v in base wn2 pp; do (timeout 300 python tools/experiments/f32_linear_variants.py $v 2>&1 | tail -30) > gpurun_out/c1/f32_$v.log 2>&1; done
(timeout 600 python bench.py --cpu-reps 1 2>&1 | tail -3) > gpurun_out/c1/bench.log 2>&1
tail -5 gpurun_out/c1/pytest.log; cat gpurun_out/c1/config5.log; cat gpurun_out/c1/f32_*.log
