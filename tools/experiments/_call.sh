cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
LANCZOSNET_HIP_LIB=tools/experiments/_variants/liblnz_lanczos_ritz_phases.so timeout 300 python tools/ritz_phase_probe.py 2>&1 | tail -22
