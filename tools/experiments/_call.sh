cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_runner_dropin.py tests/test_graph_runner_dropin.py -m gpu -q 2>&1 | tail -2
