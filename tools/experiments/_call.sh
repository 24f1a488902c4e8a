# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r04d > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/prof_r04d/final_bench.json 2> gpurun_out/prof_r04d/final_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/prof_r04d/final_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['avg_launch_ms'], d['config']['stage_ms'], d.get('parity_rel_err'))
print({k:(v.get('ms_per_step') if isinstance(v,dict) else None) for k,v in d['config'].items() if isinstance(v,dict)})
PY
