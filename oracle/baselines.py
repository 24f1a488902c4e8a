"""numpy restatement of the reference's dense message-passing baselines (TEST INFRASTRUCTURE ONLY):
GCN (model/gcn.py:66-113), DCNN (model/dcnn.py:68-121), ChebyNet (model/cheby_net.py:66-121).
Pinned against the unmodified reference classes by tests/golden/baselines.npz
(tests/golden/make_golden_baselines.py)."""
import numpy as np


def _linear(x, P, name):
  return x @ P[name + '.weight'].T + P[name + '.bias']


def baseline_forward(P, kind, node_feat, L, mask, num_layer, diffusion_dist=(), polynomial_order=0,
                     dtype=np.float64):
  """score [B,P] of GCN / DCNN / ChebyNet (eval mode).  node_feat int [B,N]; L [B,N,N,E+1];
  mask [B,N].  Message order per layer (the column blocks of filter.t.weight):
    GCN      : L[..,e] X for e = 0..E                                   (gcn.py:88-91)
    DCNN     : L[..,e] X for e = 0..E, then L_0^k X for k in diffusion_dist   (dcnn.py:82-97)
    ChebyNet : L[..,e] X for e = 1..E, then S_0 = L_0 X, S_k = 2 L_0 S_{k-1} - S_{k-2}
               (S_{-1} = X) for k < polynomial_order, then X                 (cheby_net.py:88-98)"""
  P = {k: np.asarray(v, dtype=dtype) for k, v in P.items()}
  L = np.asarray(L, dtype=dtype)
  B, N, E1 = L.shape[0], L.shape[1], L.shape[3]
  state = P['embedding.weight'][np.asarray(node_feat)]
  L0 = L[:, :, :, 0]
  for tt in range(num_layer):
    if kind == 'GCN':
      msg = [L[:, :, :, e] @ state for e in range(E1)]
    elif kind == 'DCNN':
      msg = [L[:, :, :, e] @ state for e in range(E1)]
      tmp = state
      for ii in range(1, max(diffusion_dist) + 1):
        tmp = L0 @ tmp
        if ii in diffusion_dist:
          msg.append(tmp)
    elif kind == 'ChebyNet':
      K = polynomial_order
      scale = [None] * (K + 1)
      scale[-1] = state                      # python index -1: also what kk - 2 = -1 reads
      scale[0] = L0 @ state
      for kk in range(1, K):
        scale[kk] = 2.0 * (L0 @ scale[kk - 1]) - scale[kk - 2]
      msg = [L[:, :, :, e] @ state for e in range(1, E1)] + scale
    else:
      raise ValueError(kind)
    msg = np.concatenate(msg, axis=2).reshape(B * N, -1)
    state = np.maximum(_linear(msg, P, 'filter.%d' % tt), 0).reshape(B, N, -1)
  flat = state.reshape(B * N, -1)
  y = _linear(flat, P, 'filter.%d' % num_layer)
  att = 1.0 / (1.0 + np.exp(-_linear(flat, P, 'att_func.0')))
  y = (att * y).reshape(B, N, -1)
  m = np.asarray(mask).astype(bool)
  return np.stack([y[b, m[b], :].mean(axis=0) for b in range(B)]).astype(dtype)
