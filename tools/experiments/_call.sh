cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=tools/experiments/_variants
timeout 600 python -m pytest tests/test_gpu_strips.py -m gpu -x -q 2>&1 | tail -2
timeout 200 python tools/experiments/forward_ab.py 2>&1 | tail -1
LANCZOSNET_HIP_LIB=$V/liblnz_conv_strip_ring4.so timeout 200 python tools/experiments/forward_ab.py 2>&1 | tail -1
timeout 200 python tools/experiments/forward_ab.py 2>&1 | tail -1
LANCZOSNET_HIP_LIB=$V/liblnz_conv_strip_ring4.so timeout 200 python tools/experiments/forward_ab.py 2>&1 | tail -1
