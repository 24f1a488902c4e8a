"""RCCL on the hardware that is there: the test box has ONE GPU, so the `nccl` process group is
initialised at world size 1 (a legal communicator) and the path's two exchanges — the per-step
score all-gather and the training gradient all-reduce — run on their real backend instead of the
gloo stand-in of tests/test_dist_gloo.py.  Also `bench.py --dist nccl` as the driver would launch
it (torch.distributed.run, one rank)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(script_args, timeout=900):
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  out = None
  for _ in range(3):   # a port taken between the probe and the rendezvous: try another
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', str(port)] + script_args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    if out.returncode == 0 or 'ddress already in use' not in out.stderr:
      break
  return out


def test_score_gather_and_gradient_all_reduce_on_rccl():
  out = _run([os.path.join(ROOT, 'tests', 'rccl_world1_worker.py')])
  assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
  assert 'RCCL_WORLD1_OK' in out.stdout


def test_bench_with_the_rccl_exchange_in_the_timed_loop():
  out = _run([os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--dist', 'nccl', '--steps', '8',
              '--warmup', '3', '--no-secondary', '--no-cpu-baseline'])
  assert out.returncode == 0, out.stderr[-3000:]
  lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
  assert len(lines) == 1
  d = json.loads(lines[0])
  ex = d['config']['exchange']
  assert ex['backend'] == 'nccl' and ex['world'] == 1 and ex['gathered_equals_local'] is True
  assert d['n_gpus'] == 1 and d['value'] > 0
