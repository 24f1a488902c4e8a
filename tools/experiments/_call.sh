# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/f3
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/f3/pytest.log 2>&1; tail -3 gpurun_out/f3/pytest.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/f3/bench.json 2> gpurun_out/f3/bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/f3/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['avg_launch_ms'], d['config']['stage_ms'], d.get('parity_rel_err'), d['cpu_baseline']['value'])
PY
