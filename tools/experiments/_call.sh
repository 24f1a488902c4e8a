cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/e5
timeout 600 python tools/pmc_forward_profile.py gpurun_out/e5/pmc 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/e5/bench.json 2> gpurun_out/e5/bench.err; tail -c 300 gpurun_out/e5/bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/e5/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['avg_launch_ms'], r['flops_per_launch_executed'], r['useful_row_frac'], r['tiles_per_launch'], d['config']['stage_ms'], d.get('parity_rel_err'))
PY
