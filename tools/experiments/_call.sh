cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/d2
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/d2/pytest.log 2>&1; tail -4 gpurun_out/d2/pytest.log
timeout 600 python tools/pmc_forward_profile.py gpurun_out/d2/pmc16 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/d2/bench.json 2> gpurun_out/d2/bench.err; tail -c 600 gpurun_out/d2/bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/d2/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['stage_ms'])
PY
