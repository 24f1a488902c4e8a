"""GPU parity tests of the AdaLanczosNet rows through the C ABI."""
import ast

import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _t(x):
  return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def test_ada_graph_laplacian_matches_reference():
  from lanczosnet_amd import ops
  g = load_golden('ada_lanczos.npz')
  L0 = _t(g['adj'])  # adj = (L != 0); passing the 0/1 mask itself is equivalent
  Le = ops.ada_graph_laplacian(_t(g['feat']), None, L0).cpu().numpy()
  assert rel_err(Le, g['Le']) < 1e-5
  # embedding-gather path == float-feature path
  rs = np.random.RandomState(0)
  emb = rs.randn(70, 70).astype(np.float32)
  ids = rs.randint(0, 70, size=g['adj'].shape[:2])
  a = ops.ada_graph_laplacian(_t(ids), _t(emb), L0).cpu().numpy()
  b = ops.ada_graph_laplacian(_t(emb[ids]), None, L0).cpu().numpy()
  np.testing.assert_array_equal(a, b)
  assert rel_err(a, oracle.ada_graph_laplacian(emb[ids], g['adj'])) < 1e-5


def test_ada_lanczos_layer_matches_reference_incl_quirks():
  from lanczosnet_amd import ops
  g = load_golden('ada_lanczos.npz')
  for A, Tref, Qref in ((g['A'], g['T'], g['Q']), (g['Le'], g['T2'], g['Q2'])):
    T, Q = ops.ada_lanczos_layer(_t(A), _t(g['node_mask']), _t(g['q1']), 20)
    T, Q = T.cpu().numpy(), Q.cpu().numpy()
    # quirk structure (which alpha / beta / columns / node rows are zeroed) must match exactly
    np.testing.assert_array_equal(T != 0, Tref != 0)
    np.testing.assert_array_equal(Q != 0, Qref != 0)
    # values: fp32 Lanczos amplifies summation-order noise near breakdown (the numpy oracle
    # itself differs from torch by up to 2e-4 / 2e-3 here, tests/test_oracle_golden.py)
    per_mol_T = np.abs(T - Tref).reshape(len(T), -1).max(axis=1)
    assert np.median(per_mol_T) < 1e-5 and rel_err(T, Tref) < 5e-4
    assert rel_err(Q, Qref) < 5e-3
  # 6-node fixture of SURVEY.md §A.3 (probe values from the unmodified reference, N=8 tile)
  s = load_golden('six_node.npz')
  A = np.zeros((1, 8, 8), np.float32)
  A[0, :6, :6] = s['L4']
  torch.manual_seed(1234)
  q1 = torch.randn(1, 8, 1)
  mask = np.zeros((1, 8), np.uint8); mask[0, :6] = 1
  T, Q = ops.ada_lanczos_layer(_t(A), _t(mask), q1[:, :, 0].to(DEV), 20)
  T, Q = T.cpu().numpy()[0], Q.cpu().numpy()[0]
  np.testing.assert_allclose(np.diag(T)[:6], [0.07247, 0.23152, 0.55666, 0.45030, 0.49505, 0],
                             atol=2e-5)
  np.testing.assert_allclose(np.diag(T, 1)[:5], [0.38800, 0.34134, 0.30922, 0.27092, 0.25043],
                             atol=2e-5)
  assert (Q[5:] == 0).all() and (Q[:, 5:] == 0).all()


def test_ada_t_powers_and_symmetrize():
  from lanczosnet_amd import ops
  g = load_golden('ada_lanczos.npz')
  dists = [5, 7, 10, 20, 30]
  got = ops.ada_t_powers(_t(g['T']), dists).cpu().numpy()
  ref = oracle.ada_t_powers(g['T'].astype(np.float64), dists, dtype=np.float64)
  assert rel_err(got, ref) < 1e-5
  rs = np.random.RandomState(1)
  DD = rs.randn(5, 20 * 20 * 5).astype(np.float32)
  got = ops.ada_symmetrize_filters(_t(DD), 20, 5).cpu().numpy()
  D4 = DD.reshape(5, 20, 20, 5)
  ref = (0.5 * (D4 + D4.transpose(0, 2, 1, 3))).transpose(0, 3, 1, 2)
  np.testing.assert_allclose(got, ref, rtol=0, atol=1e-7)


def _ada_model(cfg, P):
  from lanczosnet_amd.model import AdaLanczosNet
  from lanczosnet_amd.utils.arg_helper import make_model_config
  net = AdaLanczosNet(make_model_config(cfg, name='AdaLanczosNet')).eval()
  net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
  return net.to(DEV)


def test_ada_dense_filter_conv_matches_oracle_given_TQ():
  """Stage test decoupled from Lanczos sensitivity: feed the reference (T, Q) and compare the
  filter MLP -> Q DD Q^T conv -> readout against the fp64 oracle at 1e-5."""
  from lanczosnet_amd import ops
  g = load_golden('ada_full.npz')
  c = load_golden('collate_batch.npz')
  a = load_golden('ada_lanczos.npz')
  cfg = ast.literal_eval(str(g['cfg_json']))
  nb = int(g['nb'])
  P = oracle.make_ada_params(cfg, int(g['param_seed']))
  net = _ada_model(cfg, P)
  T, Q = a['T'][:nb], a['Q'][:nb]
  ref, _ = oracle.ada_lanczos_net_forward(P, cfg, c['node_feat'][:nb], c['L'][:nb],
                                          c['node_mask'][:nb], None, dtype=np.float64, TQ=(T, Q))
  K, S = cfg['num_eig_vec'], len(cfg['long_diffusion_dist'])
  with torch.no_grad():
    plan = net._plan()
    tcat = ops.ada_t_powers(_t(T), cfg['long_diffusion_dist']).view(nb, -1)
    DDp = torch.empty((cfg['num_layer'], nb, S, K, K), dtype=torch.float32, device=DEV)
    for t, seq in enumerate(net.spectral_filter):
      ops.ada_symmetrize_filters(seq(tcat), K, S, out=DDp[t])
    Lp = ops.pack_laplacian(_t(c['L'][:nb]))
    score = ops.lanczosnet_forward(plan, _t(c['node_feat'][:nb]), Lp, _t(Q), DDp,
                                   _t(c['node_mask'][:nb])).cpu().numpy()
  assert rel_err(score, ref) < 1e-5


def test_ada_lanczos_net_end_to_end_vs_reference():
  g = load_golden('ada_full.npz')
  c = load_golden('collate_batch.npz')
  cfg = ast.literal_eval(str(g['cfg_json']))
  nb = int(g['nb'])
  P = oracle.make_ada_params(cfg, int(g['param_seed']))
  net = _ada_model(cfg, P)
  real_randn = torch.randn
  q1 = torch.from_numpy(g['q1'][:, :, None].copy())
  torch.randn = lambda *a, **k: q1.clone()  # the reference draws torch.randn(B,N,1) (:161)
  try:
    with torch.no_grad():
      score = net(_t(c['node_feat'][:nb]), _t(c['L'][:nb]), mask=_t(c['node_mask'][:nb]))
  finally:
    torch.randn = real_randn
  score = score.cpu().numpy()
  e = np.abs(score - g['score']).max(axis=1) / np.abs(g['score']).max()
  assert np.median(e) < 2e-4 and e.max() < 5e-3, e  # same bar as the oracle-vs-reference pin


def test_ada_module_surface():
  from lanczosnet_amd.model import AdaLanczosNet
  from lanczosnet_amd.utils.arg_helper import make_model_config
  cfg = dict(oracle.DEFAULT_QM8_CFG, short_diffusion_dist=[1, 2, 3],
             long_diffusion_dist=[5, 7, 10, 20, 30], hidden_dim=[128], num_layer=1)
  net = AdaLanczosNet(make_model_config(cfg, name='AdaLanczosNet'))
  assert net.input_dim == 70 and net.use_reorthogonalization is True
  keys = set(net.state_dict().keys())
  assert {'embedding.weight', 'filter.0.weight', 'filter.1.bias', 'spectral_filter.0.0.weight',
          'spectral_filter.0.6.bias', 'att_func.0.weight'} <= keys
  assert net.state_dict()['spectral_filter.0.0.weight'].shape == (4096, 2000)


def test_ada_training_gradients_match_reference_autograd():
  """AdaLanczosNet loss.backward() (HIP forward + torch recomputation backward, same q1) against
  the reference's parameter-gradient statistics (tests/golden/ada_full.npz)."""
  g = load_golden('ada_full.npz')
  c = load_golden('collate_batch.npz')
  cfg = ast.literal_eval(str(g['cfg_json']))
  nb = int(g['nb'])
  P = oracle.make_ada_params(cfg, int(g['param_seed']))
  net = _ada_model(cfg, P).train()
  real_randn = torch.randn
  q1 = torch.from_numpy(g['q1'][:, :, None].copy())
  torch.randn = lambda *a, **k: q1.clone()
  try:
    score, loss = net(_t(c['node_feat'][:nb]), _t(c['L'][:nb]), label=_t(c['label'][:nb]),
                      mask=_t(c['node_mask'][:nb]))
  finally:
    torch.randn = real_randn
  assert abs(float(loss.detach()) - float(g['loss'])) < 5e-3 * abs(float(g['loss']))
  loss.backward()
  gd = dict(net.named_parameters())
  worst = 0.0
  for k, gs, ga in zip(g['gnames'], g['gsum'], g['gabs']):
    gr = gd[str(k)].grad
    assert gr is not None, k
    # fp32 Lanczos noise floor (see the forward tests): gradients agree to ~1e-3 of their mass
    e = abs(float(gr.double().abs().sum()) - float(ga)) / (float(ga) + 1e-12)
    worst = max(worst, e)
    assert e < 2e-2, (k, e)
    assert abs(float(gr.double().sum()) - float(gs)) < 2e-2 * float(ga) + 1e-9, k
  print('worst relative gradient-mass deviation %.2e' % worst)
