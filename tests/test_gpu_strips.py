"""The strip plan (lnz_plan_strips: molecules at 4-row granularity in strips of 16-row subtiles) and
the inference forward that runs on it (csrc/conv_strip.hip).  The plan is checked against a Python
restatement of the packing rule and against its invariants; the forward against the 32-row-tile
kernels on the same batch and (through the parity tests of test_gpu_parity.py, whose default launches
carry a strip plan) against the oracle."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda'
from strip_mirror import INTS, SUB, plan_strips_mirror  # noqa: E402


def _check_plan(buf, ext, n_cu):
  scap = (buf.numel() - 1) // INTS
  p = buf.cpu().numpy()
  used = int(p[scap * INTS])
  seen = np.zeros(len(ext), int)
  plan = []
  for s in range(used):
    e = p[s * INTS:(s + 1) * INTS]
    nm, sub = int(e[0]), int(e[1])
    assert 1 <= nm <= 24 and 1 <= sub <= SUB
    taken = np.zeros(16 * sub, int)
    mols = []
    for i in range(nm):
      b, st, n = (int(x) for x in e[2 + 3 * i:5 + 3 * i])
      rows = 4 if n <= 4 else (n + 3) // 4 * 4
      assert n == ext[b] and st % 4 == 0 and st + rows <= 16 * sub
      assert st // 16 + 1 >= (st + rows - 1) // 16, 'a molecule spans at most two subtiles'
      taken[st:st + rows] += 1
      seen[b] += 1
      mols.append((b, st, n))
    assert taken.max() == 1
    assert [m[1] for m in mols] == sorted(m[1] for m in mols), 'rows ascending'
    plan.append((mols, sub))
  assert (seen == 1).all()
  return plan


@pytest.mark.parametrize('B,nmin,nmax,n_cu', [(1024, 8, 26, 256), (1024, 1, 32, 256), (2048, 3, 26, 256),
                                              (5000, 8, 26, 256),
                                              (5, 1, 9, 256), (300, 20, 32, 64), (777, 1, 6, 3)])
def test_strip_plan_invariants_and_packing_rule(B, nmin, nmax, n_cu):
  from lanczosnet_amd import ops
  rs = np.random.RandomState(B + nmax)
  ext = rs.randint(nmin, nmax + 1, size=B)
  N = 32 if nmax > 26 else 26
  mask = np.zeros((B, N), np.uint8)
  for b in range(B):
    mask[b, :ext[b]] = 1
    if ext[b] > 2 and b % 7 == 0:
      mask[b, ext[b] // 2] = 0          # a hole: the extent is the LAST real node + 1
  mk = torch.from_numpy(mask).to(DEV)
  buf = ops.plan_strips(mk, n_cu=n_cu)
  plan = _check_plan(buf, ext, n_cu)
  want = plan_strips_mirror(ext, n_cu)
  assert len(plan) == len(want)
  for (got_m, got_s), (want_m, want_s) in zip(plan, want):
    assert got_s == want_s and got_m == want_m
  assert _check_plan(ops.plan_strips(mk, n_cu=n_cu), ext, n_cu) == plan   # (unused words are not written)
  # the same plan out of the fused launches' planner workgroup
  (tiles, cap), _ = ops.plan_batch(mk, True, 20, n_cu=n_cu)
  assert _check_plan(tiles.strips, ext, n_cu) == plan


def test_bench_batch_fits_five_subtiles_per_compute_unit():
  from lanczosnet_amd import ops
  from lanczosnet_amd.synthetic import draw_batch
  for seed in (0, 1):
    b = draw_batch(1024, seed=seed)
    buf = ops.plan_strips(torch.from_numpy(b['node_mask'].astype(np.uint8)).to(DEV), n_cu=256)
    plan = _check_plan(buf, b['n_nodes'], 256)
    assert len(plan) <= 256 and max(s for _, s in plan) == 5


@pytest.mark.parametrize('B,n_cu,nmin,nmax', [(1024, 256, 2, 26), (96, 7, 2, 26), (33, 256, 2, 26), (4100, 256, 8, 26),
                                              (64, 256, 1, 32), (1, 256, 5, 5), (700, 16, 27, 32)])
def test_strip_forward_matches_the_tile_kernels_and_the_oracle(B, n_cu, nmin, nmax, monkeypatch):
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import LanczosNet
  from lanczosnet_amd.synthetic import draw_batch
  from lanczosnet_amd.utils.arg_helper import make_model_config
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  P = oracle.make_lanczosnet_params(cfg, 3)
  net = LanczosNet(make_model_config(cfg)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  net = net.to(DEV)
  b = draw_batch(B, seed=11, n_min=nmin, n_max=nmax)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, 20)
  plan = net._plan()
  Lp = ops.pack_laplacian_for(plan, L)
  G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'])
  nf, mk = t(b['node_feat']), t(b['node_mask'].astype(np.uint8))
  tiles = ops.plan_tiles(mk, True, n_cu=n_cu)
  assert getattr(tiles[0], 'strips', None) is not None
  with torch.no_grad():
    s_strip, st_strip = ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles, return_state=True)
    s_fast = ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles)
    monkeypatch.setenv('LNZ_STRIPS', '0')
    s_tile, st_tile = ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles, return_state=True)
  assert torch.equal(s_strip, s_fast)
  scale = s_tile.abs().max().item()
  assert (s_strip - s_tile).abs().max().item() <= 2e-6 * scale
  # final node states of the real nodes
  real = torch.arange(32, device=DEV)[None, :] < n[:, None]
  d = (st_strip - st_tile).abs().amax(dim=2)
  assert d[real].max().item() <= 2e-6 * st_tile.abs().max().item()
  ref = oracle.lanczos_net_forward(P, cfg, b['node_feat'], L.cpu().numpy(), D.cpu().numpy(),
                                   V.cpu().numpy(), b['node_mask'], dtype=np.float64)
  got = s_strip.cpu().numpy()
  assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize('strips', ['1', '0'])
def test_a_non_finite_ritz_block_stays_with_its_molecule(strips, monkeypatch):
  """Molecules that share a strip (or a pair tile) meet in the matrix instructions, where
  0 x NaN = NaN: the kernels stage a non-finite Ritz entry as 0 (csrc/conv_tiles.hpp,
  finite_or_zero), so that a degenerate molecule — AdaLanczosNet's learned Laplacian is 0 / 0 for a
  one-node molecule whose padding carries the same embedding (model/ada_lanczos_net.py:126-129) —
  cannot change the scores of its neighbours, as it cannot in the reference's batched products."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import LanczosNet
  from lanczosnet_amd.synthetic import draw_batch
  from lanczosnet_amd.utils.arg_helper import make_model_config
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  P = oracle.make_lanczosnet_params(cfg, 3)
  net = LanczosNet(make_model_config(cfg)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  net = net.to(DEV)
  B = 93
  b = draw_batch(B, seed=5, n_min=1, n_max=12)  # N = 12: pair tiles as well as strips
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, 20)
  plan = net._plan()
  Lp = ops.pack_laplacian_for(plan, L)
  G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'])
  nf, mk = t(b['node_feat']), t(b['node_mask'].astype(np.uint8))
  tiles = ops.plan_tiles(mk, True)
  monkeypatch.setenv('LNZ_STRIPS', strips)
  bad = [7, 40, 78]
  V_nan, V_zero = V.clone(), V.clone()
  V_nan[bad[0]] = float('nan')
  V_nan[bad[1]] = float('inf')
  V_nan[bad[2], 0, 0] = float('-inf')
  V_zero[bad[0]] = 0
  V_zero[bad[1]] = 0
  V_zero[bad[2], 0, 0] = 0
  with torch.no_grad():
    s_nan = ops.lanczosnet_forward(plan, nf, Lp, V_nan, G, mk, tiling=tiles)
    s_zero = ops.lanczosnet_forward(plan, nf, Lp, V_zero, G, mk, tiling=tiles)
    s_ref = ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles)
  assert torch.equal(s_nan, s_zero)
  others = torch.ones(B, dtype=torch.bool, device=DEV)
  others[bad] = False
  assert torch.equal(s_nan[others], s_ref[others])


def test_strip_launches_from_two_threads_on_fresh_streams():
  """The strip forward, input-gradient and gain-gradient launches set their dynamic-LDS function
  attribute at EVERY launch (it is per device and `nn.DataParallel` — the reference's multi-GPU
  mechanism, runner/qm8_runner.py:62 — drives one thread per device): nothing is configured "once
  per process".  Two threads, each on a stream of its own, run the training step of the same batch
  concurrently, several times; every result must equal the single-threaded one bit for bit.  (The
  two-DEVICE variant is tests/test_gpu_multidevice.py, which needs a box with two GPUs; the CPU lint
  test_no_process_wide_mutable_state_in_the_kernels_sources keeps the guard from coming back.)"""
  import threading
  from lanczosnet_amd import ops
  from lanczosnet_amd.model import LanczosNet
  from lanczosnet_amd.synthetic import draw_batch
  from lanczosnet_amd.utils.arg_helper import make_model_config
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  P = oracle.make_lanczosnet_params(cfg, 3)
  b = draw_batch(192, seed=5)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)  # noqa: E731
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, 20)
  nf, mk, lab = t(b['node_feat']), t(b['node_mask'].astype(np.uint8)), t(b['label'])
  torch.cuda.synchronize()

  def step(net):
    net.zero_grad(set_to_none=True)
    score, loss = net(nf, L, D, V, label=lab, mask=mk)
    loss.backward()
    return [score.detach().clone()] + [p.grad.detach().clone() for p in net.parameters()]

  def make():
    net = LanczosNet(make_model_config(cfg))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    return net.to(DEV).train()

  want = step(make())
  torch.cuda.synchronize()
  out, err = {}, []

  def worker(i):
    try:
      net = make()
      s = torch.cuda.Stream()
      s.wait_stream(torch.cuda.default_stream())
      with torch.cuda.stream(s):
        for _ in range(4):
          out[i] = step(net)
      s.synchronize()
    except Exception as e:  # noqa: BLE001
      err.append(e)

  th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
  for x in th:
    x.start()
  for x in th:
    x.join()
  assert not err, err
  for i in range(2):
    assert len(out[i]) == len(want)
    for a, w in zip(out[i], want):
      assert torch.equal(a, w)
