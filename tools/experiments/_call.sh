# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_graph.py tests/test_graph_runner_dropin.py -x -q 2>&1 | tail -2
RITZ_WG_KERNELS="auto" timeout 200 python tools/experiments/ritz_wg_sizes.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
timeout 300 python tools/experiments/ritz_wg_fuzz.py 500 150 2>&1 | grep -v amdgpu.ids | tail -3
