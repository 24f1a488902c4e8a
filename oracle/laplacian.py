"""Oracle (test infrastructure): graph-Laplacian construction.

Restates reference `utils/data_helper.py:92-116` (normalize_adj) and
`utils/data_helper.py:119-166` (get_laplacian) for dense numpy inputs.
"""
import numpy as np


def _normalize_adj(A, is_sym=True, exponent=0.5):
  # utils/data_helper.py:99-116 ; r_inv with inf -> 0 guard (:106)
  rowsum = np.array(A.sum(1))
  with np.errstate(divide='ignore'):
    if is_sym:
      r_inv = np.power(rowsum, -exponent).flatten()
    else:
      r_inv = np.power(rowsum, -1.0).flatten()
  r_inv[np.isinf(r_inv)] = 0.
  R = np.diag(r_inv)
  if is_sym:
    return R.dot(A).dot(R)
  return R.dot(A)


def get_laplacian(adj, graph_laplacian_type='L1', alpha=0.5):
  """utils/data_helper.py:119-166, dense branch only."""
  assert adj.ndim == 2 and adj.shape[0] == adj.shape[1]
  I = np.eye(adj.shape[0])
  t = graph_laplacian_type
  if t == 'L1':
    return np.diag(adj.sum(axis=1).squeeze()) - adj
  if t == 'L2':
    return I - _normalize_adj(adj, True)
  if t == 'L3':
    return I - _normalize_adj(adj, False)
  if t == 'L4':
    return _normalize_adj(I + adj, True)
  if t == 'L5':
    return _normalize_adj(I + adj, False)
  if t == 'L6':
    return _normalize_adj(adj, True, exponent=alpha)
  if t == 'L7':
    return _normalize_adj(adj, False)
  raise ValueError('Unsupported Graph Laplacian!')


def laplacian_l4(adj):
  """L4 = D^-1/2 (I + A) D^-1/2, D = rowsum(I + A)  (utils/data_helper.py:155-156).

  Returns float64 like the reference (np.eye is float64)."""
  return get_laplacian(np.asarray(adj), 'L4')


def laplacian_multi_l4(adjs):
  """Per-edge-type L4 plus the simple-graph L4, laid out like the collate output
  `L[..., 0] = L_simple_4`, `L[..., 1+e] = L_multi[..., e]`
  (dataset/get_qm8_data.py:62-75, dataset/qm8.py:262).

  adjs: [n, n, E] -> [n, n, E+1] float64."""
  adjs = np.asarray(adjs)
  n, _, E = adjs.shape
  out = np.zeros((n, n, E + 1), dtype=np.float64)
  out[:, :, 0] = laplacian_l4(adjs.sum(axis=2))
  for e in range(E):
    out[:, :, 1 + e] = laplacian_l4(adjs[:, :, e])
  return out
