cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_strips.py -m gpu -x -q 2>&1 | tail -15
