"""bench.py's N > 1 path executed for real: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both on the one GPU of
the test box (LNZ_BENCH_ONE_DEVICE=1: gloo exchange, RCCL refuses two ranks on one device).  Checks
the contract of the JSON line: whole-job value over both shards, max-over-ranks timing, one line
from rank 0, async score all-gather drained.  And the plain command the driver's N = 1 run has the
shape of — `python bench.py --gpus N` with no launcher around it — spawns its N ranks itself, checks
every rank's shard against the oracle, and refuses (non-zero, nothing measured) when fewer than N
devices are visible."""
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


def _json_lines(text):
  return [ln for ln in text.splitlines() if ln.startswith('{')]


@pytest.mark.gpu
def test_bench_plain_command_spawns_its_ranks_and_checks_every_shard():
  # no torch.distributed.run, no RANK / WORLD_SIZE: bench.py is its own launcher
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
  env['LNZ_BENCH_ONE_DEVICE'] = '1'
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6',
                        '--warmup', '2', '--shard-parity', '48'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = _json_lines(out.stdout)
  assert len(lines) == 1, out.stdout[-2000:]
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['config']['global_batch'] == 2048
  ex = d['config']['exchange']
  assert ex['world'] == 2 and ex['ranks_formed'] == 2 and 're-ran itself' in ex['launcher']
  # one verified shard per rank, each its own seed, each under the bar
  shards = ex['shards']
  assert [s_['rank'] for s_ in shards] == [0, 1] and [s_['seed'] for s_ in shards] == [0, 1]
  assert all(s_['molecules_checked'] >= 40 and s_['parity_rel_err'] < 1e-5 for s_ in shards)
  assert d['parity_rel_err'] == max(s_['parity_rel_err'] for s_ in shards)


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_devices():
  # the one-GPU box: --gpus 8 must not quietly measure one GPU
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LNZ_BENCH_ONE_DEVICE')}
  import torch
  n = torch.cuda.device_count()
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n + 7), '--steps', '2',
                        '--warmup', '1'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
  assert out.returncode != 0
  assert not _json_lines(out.stdout)
  assert 'HIP device(s) visible' in out.stderr


def test_bench_without_devices_fails_loudly():
  """CPU container: no device, so even --gpus 1 measures nothing and says so (no CPU fallback)."""
  import torch
  if torch.cuda.is_available():
    pytest.skip('a GPU is visible')
  env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
  for n in ('1', '8'):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', n],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not _json_lines(out.stdout)
    assert 'HIP device(s) visible' in out.stderr


def test_bench_rejects_a_launch_whose_world_size_disagrees():
  import torch
  if torch.cuda.is_available():
    pytest.skip('checked on the CPU container (argument handling only)')
  # the device check comes first on a CPU box; with LNZ_BENCH_ONE_DEVICE it is the world check
  env = dict(os.environ, RANK='0', WORLD_SIZE='2', LOCAL_RANK='0', LNZ_BENCH_ONE_DEVICE='1')
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
  assert out.returncode != 0 and 'must agree' in out.stderr
