# scratch: the command file of the last gpurun call (tools/experiments/README.md)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
