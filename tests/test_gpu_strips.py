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


# ---- gemm_mode 1: split-precision GEMM1 (and block products) inside the strip kernel -------------
def _split_pack_mirror(W):
  """lnz_pack_rows_k8_split in numpy: [rt][32-k block][piece][lane slot][8 halves] (header)."""
  rows, cols = W.shape
  RT, NB = (rows + 31) // 32, cols // 32
  Wz = np.zeros((RT * 32, cols), np.float32)
  Wz[:rows] = W
  hi = Wz.astype(np.float16)
  lo = (Wz - hi.astype(np.float32)).astype(np.float16)
  out = np.zeros((RT, NB, 2, 128, 8), np.float16)
  for t in range(128):
    kq, wj = 2 * (t >> 6) + ((t >> 5) & 1), t & 31
    for piece, src in enumerate((hi, lo)):
      blk = src.reshape(RT, 32, NB, 32)[:, wj, :, 8 * kq:8 * kq + 8]   # [RT, NB, 8]
      out[:, :, piece, t, :] = blk
  return out.reshape(-1).view(np.float32)


def test_split_weight_pack_layout():
  from lanczosnet_amd import ops
  rs = np.random.RandomState(0)
  for rows, cols in ((128, 128), (128, 14 * 128), (96, 64), (33, 32)):
    W = (rs.randn(rows, cols) * 10.0 ** rs.randint(-3, 3, size=(rows, 1))).astype(np.float32)
    got = ops.pack_rows_k8_split(torch.from_numpy(W).to(DEV)).cpu().numpy()
    want = _split_pack_mirror(W)
    assert got.shape == want.shape and got.tobytes() == want.tobytes(), (rows, cols)
  with pytest.raises(Exception):
    ops.pack_rows_k8_split(torch.zeros((32, 48), device=DEV))   # cols not a multiple of 32


def _split_net(cfg, seed, general=False):
  from lanczosnet_amd.model import LanczosNet, LanczosNetGeneral
  from lanczosnet_amd.utils.arg_helper import make_model_config
  P = oracle.make_lanczosnet_params(cfg, seed, general=general)
  net = (LanczosNetGeneral if general else LanczosNet)(make_model_config(cfg, general=general)).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  return net.to(DEV), P


@pytest.mark.parametrize('B,n_cu,nmin,nmax,K', [(1024, 256, 2, 26, 20), (96, 7, 2, 26, 20), (33, 256, 2, 26, 12),
                                                (64, 256, 1, 32, 20), (1, 256, 5, 5, 20), (700, 16, 27, 32, 24)])
def test_split_precision_strip_forward_meets_the_parity_bar(B, n_cu, nmin, nmax, K):
  """gemm_mode = 'f16x3' on the strip plan: strips of 1..6 subtiles, single molecules, full tiles,
  other K — scores and final node states against the float64 oracle at the exact kernel's 1e-5 bar
  (measured 1e-6 .. 2e-6), next to the exact kernel's own deviation."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.synthetic import draw_batch
  cfg = dict(oracle.DEFAULT_QM8_CFG, num_eig_vec=K)
  net, P = _split_net(cfg, 3)
  b = draw_batch(B, seed=11, n_min=nmin, n_max=nmax)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, K)
  nf, mk = t(b['node_feat']), t(b['node_mask'].astype(np.uint8))
  tiles = ops.plan_tiles(mk, True, n_cu=n_cu)
  out = {}
  for mode in ('fp32', 'f16x3'):
    net.gemm_mode = mode
    plan = net._plan()
    assert plan['gemm_mode'] == (1 if mode == 'f16x3' else 0)
    Lp = ops.pack_laplacian_for(plan, L)
    assert Lp.dtype == (torch.float16 if mode == 'f16x3' else torch.float32)   # the type is the format
    G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'])
    with torch.no_grad():
      s, st = ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles, return_state=True)
      s_fast = ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling=tiles)
      s_own = ops.lanczosnet_forward(plan, nf, Lp, V, G, mk, tiling='single')  # (makes its own strip plan)
    assert torch.equal(s, s_fast)
    out[mode] = (s.cpu().numpy(), st.cpu().numpy(), s_own.cpu().numpy())
  ref, rst = oracle.lanczos_net_forward(P, cfg, b['node_feat'], L.cpu().numpy(), D.cpu().numpy(),
                                        V.cpu().numpy(), b['node_mask'], dtype=np.float64, return_state=True)
  real = np.arange(32)[None, :] < b['n_nodes'][:, None]
  N = b['node_mask'].shape[1]
  for mode, (s, st, s_own) in out.items():
    e_s = np.abs(s - ref).max() / np.abs(ref).max()
    e_st = np.abs(st[:, :N][real[:, :N]] - rst[real[:, :N]]).max() / np.abs(rst).max()
    print('%s: scores %.2e, node states %.2e of the float64 oracle' % (mode, e_s, e_st))
    assert e_s <= 1e-5 and e_st <= 1e-5, mode
    assert np.abs(s_own - ref).max() <= 1e-5 * np.abs(ref).max()


def test_split_precision_strip_forward_float_features_and_no_long_scales():
  """LanczosNetGeneral-style float features (input width 10, zero-padded to the kernel's 128 columns)
  and a model without long-diffusion scales (no eigen-space block at all)."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.synthetic import draw_batch
  from graph_fixture import GRAPH_CFG
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
  rs = np.random.RandomState(2)
  for cfg, general in ((dict(GRAPH_CFG, input_dim=10), True),
                       (dict(oracle.DEFAULT_QM8_CFG, long_diffusion_dist=[]), False)):
    net, P = _split_net(cfg, 4, general=general)
    net.gemm_mode = 'f16x3'
    B = 50
    b = draw_batch(B, seed=3, n_min=4, n_max=30, num_bond_type=cfg['num_bond_type'])
    n = t(b['n_nodes'])
    L = ops.laplacian_l4(t(b['adjs']), n)
    K = cfg['num_eig_vec']
    D, V = ops.lanczos_ritz(L[..., 0], n, K)
    feat = rs.randn(B, b['node_mask'].shape[1], 10).astype(np.float32) if general else b['node_feat']
    with torch.no_grad():
      got = net(t(feat), L, D, V, mask=t(b['node_mask'])).cpu().numpy()
    assert net._plan()['gemm_mode'] == 1 and net._plan()['din0'] == 128
    ref = oracle.lanczos_net_forward(P, cfg, feat, L.cpu().numpy(), D.cpu().numpy(), V.cpu().numpy(),
                                     b['node_mask'], dtype=np.float64, general=general)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), general


def test_split_precision_needs_the_strip_plan_and_says_so():
  """gemm_mode 1 through the raw entry point without a strip plan, and a model with short-diffusion
  channels (the split-precision instantiation has none): errors, not silent fallbacks."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.synthetic import draw_batch
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  net, _ = _split_net(cfg, 3)
  net.gemm_mode = 'f16x3'
  plan = net._plan()
  b = draw_batch(8, seed=1)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, 20)
  Lp = ops.pack_laplacian_for(plan, L)
  G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'])
  ext = ops._ext()
  consts = ([int(x) for x in plan['w_off'][:7]], [int(x) for x in plan['b_off'][:7]],
            [7, plan['din0'], 128, plan['dout'], plan['n_long'], plan['n_edge'], 0, 1], [])
  mk = t(b['node_mask'].astype(np.uint8))
  with pytest.raises(RuntimeError, match='strip plan'):
    ext.forward(t(b['node_feat']), plan['embedding'], Lp, None, V, G, mk, plan['Wp'], plan['bias'],
                consts[0], consts[1], plan['Wp_head'], plan['bias_head'], None, 0, consts[2], consts[3], None, 0)
  cfg_s = dict(cfg, short_diffusion_dist=[1, 2])
  net_s, _ = _split_net(cfg_s, 3)
  net_s.gemm_mode = 'f16x3'
  with pytest.raises(NotImplementedError, match='strip kernel'):
    net_s._plan()


def test_laplacian_pack_conversion_alone_and_under_the_gains_launch():
  """lnz_split_laplacian_pack_to (every fragment float4 -> 4 fp16 hi | 4 lo pieces, into a NEW float16
  tensor: the element type carries the format) against numpy, and the same conversion riding along
  with the gains launch (lnz_spectral_gains_rows_split_to): same bytes, the gains and the fp32 pack
  untouched; a clone / view of the float16 pack is still refused by an exact-fp32 plan, an fp32 pack
  handed to the split plan is converted on the way; the in-place C entry gives the same bytes."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.synthetic import draw_batch
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  net, _ = _split_net(cfg, 3)
  b = draw_batch(77, seed=4)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  D, V = ops.lanczos_ritz(L[..., 0], n, 20)
  Lp = ops.pack_laplacian(L)
  before = Lp.clone()
  raw = Lp.cpu().numpy().reshape(-1, 4)
  hi = raw.astype(np.float16)
  lo = (raw - hi.astype(np.float32)).astype(np.float16)
  want = np.concatenate([hi, lo], axis=1).reshape(-1)
  alone = ops.split_laplacian_pack(Lp)
  assert alone.dtype == torch.float16 and alone.numel() == 2 * Lp.numel() and alone.data_ptr() != Lp.data_ptr()
  assert alone.cpu().numpy().reshape(-1).tobytes() == want.tobytes()
  assert torch.equal(Lp, before)                         # the fp32 pack is what it was
  assert ops.split_laplacian_pack(alone) is alone        # a float16 pack is already converted
  plan32 = net._plan()
  G0 = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan32['mlp_pack'])
  G1, ride = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan32['mlp_pack'], split_pack=Lp)
  assert torch.equal(G0, G1) and torch.equal(Lp, before)
  assert ride.dtype == torch.float16 and ride.cpu().numpy().reshape(-1).tobytes() == want.tobytes()
  assert torch.equal(ride.ident, Lp.ident)
  nf, mk = t(b['node_feat']), t(b['node_mask'].astype(np.uint8))
  for converted in (ride, ride.clone(), ride.detach().view(-1).view(ride.shape)):   # the format is in the type
    with pytest.raises(RuntimeError, match='float16 form'):
      ops.lanczosnet_forward(plan32, nf, converted, V, G0, mk)
  net.gemm_mode = 'f16x3'
  plan16 = net._plan()
  s_typed = ops.lanczosnet_forward(plan16, nf, ride, V, G0, mk)
  s_onfly = ops.lanczosnet_forward(plan16, nf, Lp, V, G0, mk)   # an fp32 pack: converted on the way, not in place
  assert torch.equal(s_typed, s_onfly) and torch.equal(Lp, before)
  # a clone keeps the format (its type) and loses only the identity-channel bits: every channel then
  # goes through its fragments — the same scores to rounding
  s_clone = ops.lanczosnet_forward(plan16, nf, ride.clone(), V, G0, mk)
  assert float((s_clone - s_typed).abs().max()) <= 1e-5 * float(s_typed.abs().max())
  # the C ABI's in-place entry (a C host that keeps its own books): the same bytes
  inplace = Lp.clone()
  ops._abi().split_laplacian_pack(inplace, inplace.numel())
  assert inplace.view(torch.float16).cpu().numpy().reshape(-1).tobytes() == want.tobytes()


@pytest.mark.parametrize('B,nmin,nmax', [(1024, 8, 26), (37, 1, 32), (5, 3, 9)])
def test_message_pass_on_strips_matches_the_definition(B, nmin, nmax):
  """lnz_lanczosnet_messages (csrc/conv_strip.hip strip_messages; the 32-row-tile form was retired in
  r05 after agreeing with this one to 2e-6) against the definition msg[:, c] = M_c X_l in float64
  (long scales: V diag(g) V^T), layer 0 (input width 64) and a hidden layer, compact and padded row
  numbering."""
  from lanczosnet_amd import ops
  from lanczosnet_amd.synthetic import draw_batch
  cfg = dict(oracle.DEFAULT_QM8_CFG)
  net, _ = _split_net(cfg, 6)
  b = draw_batch(B, seed=9, n_min=nmin, n_max=nmax)
  t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
  n = t(b['n_nodes'])
  L = ops.laplacian_l4(t(b['adjs']), n)
  K = 20
  D, V = ops.lanczos_ritz(L[..., 0], n, K)
  plan = net._plan()
  mk = t(b['node_mask'].astype(np.uint8))
  Lp, tiles, rows = ops.pack_and_plan(plan, L, mk, K)
  assert getattr(tiles[0], 'strips', None) is not None
  G = ops.spectral_gains(D, cfg['long_diffusion_dist'], cfg['num_layer'], plan['mlp_pack'], rows=rows)
  Lnum, N = cfg['num_layer'], b['node_mask'].shape[1]
  act = torch.zeros((Lnum, B, 32, 128), device=DEV)
  with torch.no_grad():
    ops.lanczosnet_forward(plan, t(b['node_feat']), Lp, V, G, mk, tiling=tiles, act_out=act)
  x0 = torch.zeros((B, 32, plan['din0']), device=DEV)
  x0[:, :N, :cfg['input_dim']] = net.embedding.weight.detach()[t(b['node_feat'])]
  n_mol = n.long()
  row_off = (torch.cumsum(n_mol, 0) - n_mol).contiguous()
  R_tot = int(n_mol.sum())
  Cn = plan['n_long'] + plan['n_edge']
  Ld, Vd, Gd = L.double(), V.double(), G.double()
  for layer in (0, 3):
    d = plan['din0'] if layer == 0 else 128
    X = (x0 if layer == 0 else act[layer - 1])[:, :N].double()                    # [B, N, d]
    want = []
    for s in range(plan['n_long']):
      want.append(Vd @ (Gd[layer, :, s, :, None] * (Vd.transpose(1, 2) @ X)))
    for e in range(plan['n_edge']):
      want.append(Ld[:, :, :, e] @ X)
    want = torch.stack(want, dim=2)                                                # [B, N, C, d]
    real = torch.arange(N, device=DEV)[None, :] < n[:, None]
    for strips in ('1',):
      msg_c = torch.full((R_tot, Cn * d), float('nan'), device=DEV)
      ops.lanczosnet_messages(plan, Lp, V, G, mk, act, x0, layer, msg_c, tiles, row_off=row_off)
      msg_p = torch.zeros((B * 32, Cn * d), device=DEV)
      ops.lanczosnet_messages(plan, Lp, V, G, mk, act, x0, layer, msg_p, tiles)
      assert torch.isfinite(msg_c).all()          # every compact row is written
      got = msg_c.double().view(R_tot, Cn, d)
      ref = want[real]                             # rows in (molecule, node) order = the compact numbering
      assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), (layer, strips)
      padded = msg_p.view(B, 32, Cn, d)[:, :N][real].double()
      assert float((padded - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), (layer, strips)
