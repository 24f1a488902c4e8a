# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 500 python tools/experiments/ritz_wg_fuzz.py 0 500 2>&1 | grep -v amdgpu.ids | tail -8
