# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/bench_ritz_wg.py 2>/dev/null > gpurun_out/ritz_wg.jsonl
python -c "
import sys, json
for l in open('gpurun_out/ritz_wg.jsonl'):
    if l.startswith('{'):
        d = json.loads(l); print(d['case'], d['B'], d['N'], {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"
