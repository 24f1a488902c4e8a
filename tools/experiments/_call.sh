cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/d3
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/d3/pytest.log 2>&1; tail -4 gpurun_out/d3/pytest.log
for v in 0 1 0 1; do LNZ_FORWARD16=$v timeout 300 python tools/bench_train_step.py 2>&1 | tail -2; done
