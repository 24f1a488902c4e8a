cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
LANCZOSNET_HIP_LIB=tools/experiments/_variants/liblnz_lanczos_ritz_wg_phases.so timeout 300 python tools/ritz_wg_phase_probe.py 2>&1 | tail -10
timeout 300 python tools/bench_ritz_wg.py 2>&1 | tail -6
