"""Helpers shared by the CPU and GPU tests of the reference's graph configuration
(tests/golden/graph_config.npz, written by tests/golden/make_golden_graph.py from the unmodified
reference: dataset/get_graph_data.py -> dataset/graph_data.py -> model/lanczos_net_general.py)."""
import numpy as np

from conftest import load_golden

GRAPH_CFG = dict(  # config/graph_lanczos_net.yaml:9-27
    num_bond_type=1, short_diffusion_dist=[], long_diffusion_dist=[1, 2, 3, 5, 7, 10, 20, 30],
    num_eig_vec=20, spectral_filter_kind='MLP', input_dim=10, hidden_dim=[128] * 7, output_dim=2,
    num_layer=7, num_atom=0)


def unpack_adj(bits, n):
  iu = np.triu_indices(n, 1)
  v = np.unpackbits(bits)[:len(iu[0])]
  a = np.zeros((n, n), np.float32)
  a[iu] = v
  return a + a.T


def load_split(split):
  """Raw graphs + the reference's collated outputs of one split ('train': B = 10, 'test': B = 64)."""
  g = load_golden('graph_config.npz')
  t = split + '_'
  n = g[t + 'n_nodes']
  off = g[t + 'adj_off']
  roff = np.cumsum([0] + [int(x) for x in n])
  items = []
  for b in range(len(n)):
    items.append(dict(adjs=unpack_adj(g[t + 'adj_bits'][off[b]:off[b + 1]], int(n[b]))[:, :, None],
                      node_feat=g[t + 'node_feat'][roff[b]:roff[b + 1]],
                      label=g[t + 'label'][b:b + 1].astype(np.float64)))
  ref = {k: g[t + k] for k in ('D', 'V', 'D_full', 'score', 'label', 'n_nodes')}
  ref['loss'] = float(g[t + 'loss'])
  if split == 'train':
    ref['L0'] = g['train_L0']
  return items, ref, int(g['param_seed']), float(g['param_checksum'])


def pad_batch(items):
  n = np.array([it['node_feat'].shape[0] for it in items], np.int32)
  B, N = len(items), int(n.max())
  adjs = np.zeros((B, N, N, 1), np.float32)
  X = np.zeros((B, N, items[0]['node_feat'].shape[1]), np.float32)
  mask = np.zeros((B, N), np.uint8)
  for b, it in enumerate(items):
    adjs[b, :n[b], :n[b]] = it['adjs']
    X[b, :n[b]] = it['node_feat']
    mask[b, :n[b]] = 1
  return adjs, X, mask, n


def check_ritz(D, V, Dref, Vref, n_nodes, Dfull, K, powers=(1, 5, 30), tol_d=1e-6, tol_p=1e-5):
  """Sorted Ritz values within tol_d (abs), spectral projectors V diag(D^p) V^T within tol_p
  (relative to the projector's largest entry) — SURVEY.md §8(c); graphs whose top-K cut splits a
  degenerate |lambda| cluster are basis dependent and skipped.  Returns (worst D, worst projector,
  number of graphs checked)."""
  import oracle
  wd = wp = 0.0
  checked = 0
  for b in range(D.shape[0]):
    n = int(n_nodes[b])
    if oracle.degenerate_cut(Dfull[b][:n], K):
      continue
    checked += 1
    wd = max(wd, float(np.abs(D[b] - Dref[b]).max()))
    assert np.abs(D[b] - Dref[b]).max() < tol_d, (b, n, np.abs(D[b] - Dref[b]).max())
    assert (V[b, n:] == 0).all() and (V[b, :, min(n, K):] == 0).all()
    for p in powers:
      a = oracle.spectral_projector(D[b], V[b], p)
      r = oracle.spectral_projector(Dref[b], Vref[b], p)
      e = float(np.abs(a - r).max() / np.abs(r).max())
      wp = max(wp, e)
      assert e < tol_p, (b, n, p, e)
  return wd, wp, checked
