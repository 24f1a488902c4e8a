cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c8
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/c8/pytest.log 2>&1
(timeout 600 python bench.py --cpu-reps 1 2>&1 | tail -2) > gpurun_out/c8/bench.log 2>&1
tail -30 gpurun_out/c8/pytest.log; tail -c 1500 gpurun_out/c8/bench.log
