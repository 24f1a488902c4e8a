cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/fuzz_backward.py 15 16 2>&1 | tail -2
LNZ_STRIPS=0 timeout 300 python tools/fuzz_backward.py 15 16 2>&1 | tail -2
LNZ_FORWARD16=0 timeout 300 python tools/fuzz_backward.py 15 16 2>&1 | tail -2
LNZ_FORWARD16=0 timeout 300 python tools/fuzz_backward.py 60 140 2>&1 | tail -4
timeout 300 python tools/fuzz_backward.py 60 140 2>&1 | tail -4
