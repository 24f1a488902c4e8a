# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench_head.json 2> gpurun_out/bench_head.err; tail -c 300 gpurun_out/bench_head.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_head.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
c = d['config']
print(c['graph_config_mode']['stage_ms'], c['graph_config_mode']['ms_per_batch'])
print(c['large_graph_conv_mode'].get('ms') or {k: v for k, v in c['large_graph_conv_mode'].items() if 'ms' in k})
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
