# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_strips.py -q -x -k "non_finite" 2>&1 | tail -5
timeout 600 python tools/experiments/ada_strip_one.py 32 2>&1 | grep -v "nan molecules" | tail -8
timeout 900 python tools/experiments/ada_strip_fuzz.py 0 100 2>&1 | tail -4
