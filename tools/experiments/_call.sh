cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=tools/experiments/_variants
timeout 200 python tools/experiments/forward_ab.py 2>&1 | tail -1
for v in s1 s2 s3; do LANCZOSNET_HIP_LIB=$V/liblnz_conv_strip_$v.so timeout 200 python tools/experiments/forward_ab.py 2>&1 | tail -1; done
timeout 200 python tools/experiments/forward_ab.py 2>&1 | tail -1
