cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/d7
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/d7/pytest.log 2>&1; tail -12 gpurun_out/d7/pytest.log
