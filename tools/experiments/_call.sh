cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_graph.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1; do LNZ_STRIPS=$v timeout 300 python tools/bench_train_step.py 2>&1 | tail -1 | cut -c1-300; done
