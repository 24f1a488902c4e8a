cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/e7
timeout 600 python tools/pmc_forward_profile.py gpurun_out/e7/pmc 2>&1 | tail -1
