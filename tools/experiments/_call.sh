# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/f4
timeout 600 python bench.py --sweep --no-secondary --no-cpu-baseline > gpurun_out/f4/sweep.json 2> gpurun_out/f4/sweep.err; tail -c 300 gpurun_out/f4/sweep.err
python - <<PY
import json
d=json.loads(open('gpurun_out/f4/sweep.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for e in d['config']['forward_batch_sweep']: print(e)
PY
