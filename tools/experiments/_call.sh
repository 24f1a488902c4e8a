cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c11
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -x -k "train or grad or backward" 2>&1 | tail -4) > gpurun_out/c11/pytest.log 2>&1
(timeout 300 python tools/bench_train_step.py 2>&1 | tail -1) > gpurun_out/c11/train.log 2>&1
tail -3 gpurun_out/c11/pytest.log; cat gpurun_out/c11/train.log
