cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
LANCZOSNET_HIP_LIB=tools/experiments/_variants/liblnz_lanczos_ritz_always.so timeout 600 python tools/experiments/ritz_quality.py 24 2>&1 | tail -1
