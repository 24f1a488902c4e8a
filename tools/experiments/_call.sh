# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "large or graph or general" 2>&1 | tail -2
timeout 300 python tools/experiments/graph_leg.py 30 2>&1 | grep -v amdgpu.ids | tail -1
