cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c13
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_train_graph.py -m gpu -q -x -s 2>&1 | tail -40) > gpurun_out/c13/pytest.log 2>&1
cat gpurun_out/c13/pytest.log
