# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_graph.py -x -q -k "boundaries" 2>&1 | grep -v "^$" | tail -30
