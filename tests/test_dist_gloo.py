"""world_size-2 (and 3, uneven) `gloo` tests of the N > 1 path on CPU: contiguous batch shards,
score all-gather, size-weighted loss.  The per-shard forward is the numpy oracle here (tests may
use it); on the GPU box the same driver code wraps the HIP forward (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from lanczosnet_amd import dist as lnz_dist

CFG = dict(num_atom=11, num_bond_type=2, short_diffusion_dist=[], long_diffusion_dist=[1, 3],
           num_eig_vec=6, spectral_filter_kind='MLP', input_dim=8, hidden_dim=[16, 16],
           output_dim=4, num_layer=2)


def _make_batch(B=7):
  from lanczosnet_amd.synthetic import draw_batch
  b = draw_batch(B, seed=3, n_min=3, n_max=9, num_atom=11, num_bond_type=2, num_label=4)
  N = b['node_mask'].shape[1]
  L = np.zeros((B, N, N, 3), np.float32)
  Dl, Vl = [], []
  for i in range(B):
    n = int(b['n_nodes'][i])
    L[i, :n, :n] = oracle.laplacian_multi_l4(b['adjs'][i, :n, :n])
    e, V, _ = oracle.graph_laplacian_eigs(b['adjs'][i, :n, :n].sum(axis=2),
                                          graph_laplacian_type='L4')
    Dl.append(e)
    Vl.append(V)
  D, V = oracle.collate_eigs(Dl, Vl, N, 6)
  return dict(node_feat=torch.from_numpy(b['node_feat']), L=torch.from_numpy(L),
              D=torch.from_numpy(D), V=torch.from_numpy(V),
              node_mask=torch.from_numpy(b['node_mask']), label=torch.from_numpy(b['label']))


def _oracle_forward(P):
  def fn(shard):
    s = oracle.lanczos_net_forward(P, CFG, shard['node_feat'].numpy(), shard['L'].numpy(),
                                   shard['D'].numpy(), shard['V'].numpy(),
                                   shard['node_mask'].numpy())
    return torch.from_numpy(s)
  return fn


def _worker(rank, world, port, out_dir):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    P = oracle.make_lanczosnet_params(CFG, 3)
    batch = _make_batch()
    n = batch['L'].shape[0]
    full, loss = lnz_dist.forward_sharded(_oracle_forward(P), batch, n)
    ref = _oracle_forward(P)(batch)
    ref_loss = torch.mean((ref - batch['label']) ** 2)
    ok = bool(torch.equal(full, ref)) and abs(float(loss) - float(ref_loss)) < 1e-6
    lo, hi = lnz_dist.shard_bounds(n, rank, world)
    np.save(os.path.join(out_dir, 'r%d.npy' % rank), np.array([ok, lo, hi], dtype=np.int64))
  finally:
    dist.destroy_process_group()


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _spawn(worker, world, out_dir, attempts=3):
  """mp.spawn on a fresh port; a rendezvous that fails (the port was taken between the probe and
  the bind, or a loaded host missed the store timeout) is retried on another port."""
  last = None
  for _ in range(attempts):
    try:
      mp.spawn(worker, args=(world, _free_port(), out_dir), nprocs=world, join=True)
      return
    except Exception as e:  # ProcessRaisedException / ProcessExitedException
      msg = str(e)
      if not any(k in msg for k in ('Address already in use', 'address already in use', 'timed out',
                                    'Timed out', 'Connection refused', 'Connection reset',
                                    'DistNetworkError', 'DistStoreError')):
        raise
      last = e
  raise last


def _grad_worker(rank, world, port, out_dir):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    x, y = torch.randn(11, 5), torch.randn(11, 3)
    ref = torch.autograd.grad(torch.nn.functional.mse_loss(net(x), y), list(net.parameters()))
    lo, hi = lnz_dist.shard_bounds(11, rank, world)  # uneven shards: 11 rows over 2 or 3 ranks
    torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]).backward()
    # 48-byte buckets: the 7x5 weight travels alone, the small tensors share buckets
    lnz_dist.all_reduce_gradients(net.parameters(), hi - lo, bucket_bytes=48 if rank >= 0 else None)
    ok = all(torch.allclose(p.grad, g, rtol=1e-5, atol=1e-7) for p, g in zip(net.parameters(), ref))
    np.save(os.path.join(out_dir, 'g%d.npy' % rank), np.array([ok], dtype=np.int64))
  finally:
    dist.destroy_process_group()


def _gather_worker(rank, world, port, out_dir):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    rows, width, steps = 5, 4, 7
    g = lnz_dist.AsyncScoreGather(rows, width, 'cpu', depth=2)
    ok = True
    tickets = []
    for k in range(steps):
      local = torch.full((rows, width), float(100 * k + rank))
      tickets.append(g.submit(local))
      if k >= 1:  # the previous step's result is still held (depth 2)
        prev = g.result(tickets[k - 1])
        want = torch.cat([torch.full((rows, width), float(100 * (k - 1) + r)) for r in range(world)])
        ok = ok and bool(torch.equal(prev, want))
    g.drain()
    last = g.result(tickets[-1])
    want = torch.cat([torch.full((rows, width), float(100 * (steps - 1) + r)) for r in range(world)])
    ok = ok and bool(torch.equal(last, want))
    try:
      g.result(tickets[0])  # long overwritten: must refuse
      ok = False
    except AssertionError:
      pass
    np.save(os.path.join(out_dir, 'a%d.npy' % rank), np.array([ok], dtype=np.int64))
  finally:
    dist.destroy_process_group()


def _subgroup_worker(rank, world, port, out_dir):
  """forward_sharded / all_reduce_gradients on a SUB-group (ranks 1, 2 of 3) and with a parameter
  that has no gradient on one rank only: the slices follow the group's rank, and the bucket layout
  does not depend on which gradients exist."""
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    grp = dist.new_group([1, 2])
    ok = True
    if rank in (1, 2):
      P = oracle.make_lanczosnet_params(CFG, 3)
      batch = _make_batch()
      n = batch['L'].shape[0]
      full, loss = lnz_dist.forward_sharded(_oracle_forward(P), batch, n, group=grp)
      ref = _oracle_forward(P)(batch)
      ok = bool(torch.equal(full, ref)) and \
          abs(float(loss) - float(torch.mean((ref - batch['label']) ** 2))) < 1e-6
      # gradients: rank 2's shard never touches the second head -> its grad is None there
      torch.manual_seed(0)
      a, b2 = torch.nn.Linear(5, 3), torch.nn.Linear(5, 3)
      x, y = torch.randn(8, 5), torch.randn(8, 3)
      gr = dist.get_rank(grp)
      lo, hi = lnz_dist.shard_bounds(8, gr, 2)
      out = a(x[lo:hi]) + (b2(x[lo:hi]) if gr == 0 else 0.0)
      torch.nn.functional.mse_loss(out, y[lo:hi]).backward()
      unused = torch.nn.Linear(5, 3)   # no rank's loss touches it: stays without a gradient
      params = list(a.parameters()) + list(b2.parameters()) + list(unused.parameters())
      assert (b2.weight.grad is None) == (gr == 1)
      lnz_dist.all_reduce_gradients(params, hi - lo, group=grp, bucket_bytes=64)
      ok = ok and unused.weight.grad is None and unused.bias.grad is None
      params = params[:4]
      # reference: the same two-shard loss in one process
      a2, b3 = torch.nn.Linear(5, 3), torch.nn.Linear(5, 3)
      a2.load_state_dict(a.state_dict())
      b3.load_state_dict(b2.state_dict())
      l0 = torch.nn.functional.mse_loss(a2(x[:4]) + b3(x[:4]), y[:4], reduction='sum')
      l1 = torch.nn.functional.mse_loss(a2(x[4:]), y[4:], reduction='sum')
      ((l0 + l1) / y.numel()).backward()
      for p_, q_ in zip(params, list(a2.parameters()) + list(b3.parameters())):
        ok = ok and torch.allclose(p_.grad, q_.grad, rtol=1e-5, atol=1e-7)
    np.save(os.path.join(out_dir, 's%d.npy' % rank), np.array([ok], dtype=np.int64))
  finally:
    dist.destroy_process_group()


def test_subgroup_sharding_and_missing_gradients(tmp_path):
  _spawn(_subgroup_worker, 3, str(tmp_path))
  for r in range(3):
    assert np.load(os.path.join(str(tmp_path), 's%d.npy' % r))[0] == 1


def test_helpers_without_process_group():
  """Single process, torch.distributed not initialised: the helpers degrade to the local result
  instead of raising."""
  assert not dist.is_initialized()
  P = oracle.make_lanczosnet_params(CFG, 3)
  batch = _make_batch()
  full, loss = lnz_dist.forward_sharded(_oracle_forward(P), batch, batch['L'].shape[0])
  ref = _oracle_forward(P)(batch)
  assert torch.equal(full, ref)
  assert abs(float(loss) - float(torch.mean((ref - batch['label']) ** 2))) < 1e-6
  # all_reduce_gradients leaves a single process's gradients exactly as they are: no zeros for
  # parameters the backward did not touch (the optimizer skips .grad = None, as the reference's does)
  lin, idle = torch.nn.Linear(4, 2), torch.nn.Linear(4, 2)
  lin(torch.ones(3, 4)).sum().backward()
  g0 = lin.weight.grad.clone()
  lnz_dist.all_reduce_gradients(list(lin.parameters()) + list(idle.parameters()), 3)
  assert torch.equal(lin.weight.grad, g0) and idle.weight.grad is None


@pytest.mark.parametrize('world', [2, 3])
def test_async_score_gather_streams_batches(world, tmp_path):
  """AsyncScoreGather (bench.py's per-step exchange): every submitted shard score arrives in
  rank order in its own result buffer, `depth` steps stay readable, older tickets are refused."""
  _spawn(_gather_worker, world, str(tmp_path))
  for r in range(world):
    assert np.load(os.path.join(str(tmp_path), 'a%d.npy' % r))[0] == 1


@pytest.mark.parametrize('world', [2, 3])
def test_gradient_all_reduce_is_the_full_batch_gradient(world, tmp_path):
  """Shard-size-weighted flat-bucket all-reduce == gradient of the unsharded mean loss."""
  _spawn(_grad_worker, world, str(tmp_path))
  for r in range(world):
    assert np.load(os.path.join(str(tmp_path), 'g%d.npy' % r))[0] == 1


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_forward_matches_single_process(world, tmp_path):
  _spawn(_worker, world, str(tmp_path))
  covered = []
  for r in range(world):
    ok, lo, hi = np.load(tmp_path / ('r%d.npy' % r))
    assert ok == 1
    covered += list(range(lo, hi))
  assert covered == list(range(7))  # shards tile the batch exactly once, in order


def test_shard_bounds_properties():
  for n in (0, 1, 7, 1024, 8191):
    for world in (1, 2, 3, 8):
      b = [lnz_dist.shard_bounds(n, r, world) for r in range(world)]
      assert b[0][0] == 0 and b[-1][1] == n
      assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
      sizes = [hi - lo for lo, hi in b]
      assert max(sizes) - min(sizes) <= 1
