"""bench.py's N > 1 path executed for real: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both on the one GPU of
the test box (LNZ_BENCH_ONE_DEVICE=1: gloo exchange, RCCL refuses two ranks on one device).  Checks
the contract of the JSON line: whole-job value over both shards, max-over-ranks timing, one line
from rank 0, async score all-gather drained."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_one_json_line():
  env = dict(os.environ, LNZ_BENCH_ONE_DEVICE='1', MASTER_ADDR='127.0.0.1')
  for attempt in range(3):   # a port taken between the probe and the rendezvous: try another
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
           '--gpus', '2', '--steps', '6', '--warmup', '2']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if out.returncode == 0 or 'ddress already in use' not in out.stderr:
      break
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
  assert len(lines) == 1, out.stdout[-2000:]          # rank 0 only
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['steps'] == 6 and d['warmup'] == 2
  assert d['scaling'] == 'weak' and d['config']['global_batch'] == 2048
  assert d['unit'] == 'molecules/s' and d['higher_is_better'] is True
  # whole-job aggregate: 2 shards of 1024 molecules per step over the max-over-ranks step time
  assert abs(d['value'] - 2048 / (d['ms_per_step'] * 1e-3)) <= 1e-3 * d['value']
  assert d['roofline']['tiles_per_launch'] > 0 and 'cpu_baseline' not in d
  # the line explains itself: world size, backend and every rank's own step time
  ex = d['config']['exchange']
  assert ex['world'] == 2 and ex['gathered_equals_local'] is True
  assert len(ex['ms_per_step_per_rank']) == 2
  assert abs(max(ex['ms_per_step_per_rank']) - d['ms_per_step']) <= 1e-3 * d['ms_per_step'] + 1e-3
