cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 1 0 1 0; do LNZ_PREP_SPLIT=$v timeout 300 python tools/bench_prep.py 100 2>&1 | tail -1 | sed "s/^/split=$v /"; done
for v in 1 0; do LNZ_PREP_SPLIT=$v timeout 300 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$v', d['value'], d['ms_per_step'], d['config']['stage_ms'])"; done
