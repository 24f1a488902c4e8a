cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_strips.py -m gpu -x -q 2>&1 | tail -8
for B in 4096 16384; do for v in 0 1; do PROBE_B=$B LNZ_STRIPS=$v timeout 300 python tools/experiments/forward_ab.py 2>&1 | tail -1 | sed "s/^/B=$B strips=$v /"; done; done
