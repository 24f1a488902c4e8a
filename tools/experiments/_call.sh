cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 1 0 1; do LNZ_PREP_MERGED=$v timeout 300 python tools/bench_prep.py 100 2>&1 | tail -1 | sed "s/^/merged=$v /"; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_strips.py -m gpu -x -q 2>&1 | tail -2
