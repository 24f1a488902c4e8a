cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/e4
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/e4/pytest.log 2>&1; tail -6 gpurun_out/e4/pytest.log
