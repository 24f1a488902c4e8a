"""Oracle (test infrastructure): AdaLanczosNet's in-model Lanczos layer and learned Laplacian.

Restates reference `model/ada_lanczos_net.py:139-247` (`_lanczos_layer`, with the
three quirks of SURVEY.md F6 / §A.3 reproduced on purpose) and `:101-137`
(`_get_graph_laplacian`).  The start vector q1 is an argument (the reference draws
`torch.randn(B,N,1)` on the CPU generator at :161).
"""
import numpy as np

EPS = float(np.finfo(np.float32).eps)  # model/ada_lanczos_net.py:8


def ada_lanczos_layer(A, mask, q1, num_eig_vec, use_reorth=True, dtype=np.float32,
                      return_raw_betas=False):
  """A: [B,N,N] sym, mask: [B,N] (0/1) or None, q1: [B,N] raw start vector (before masking).

  Returns T [B,K,K], Q [B,N,K] (+ the RAW beta_1..beta_T of every step, [B,T], before the
  breakdown mask is applied — what the parity protocol classifies molecules by)."""
  A = np.asarray(A, dtype=dtype)
  B, N = A.shape[0], A.shape[1]
  K = num_eig_vec
  T_it = min(N, K)
  eps = dtype(EPS)

  Q = [None] * (T_it + 2)
  alpha = [None] * (T_it + 1)
  beta = [None] * (T_it + 1)
  beta[0] = np.zeros((B, 1, 1), dtype=dtype)
  Q[0] = np.zeros((B, N, 1), dtype=dtype)
  Q[1] = np.asarray(q1, dtype=dtype).reshape(B, N, 1).copy()
  m = None
  if mask is not None:
    m = np.asarray(mask).astype(dtype)[:, :, None]
    Q[1] = Q[1] * m  # :163-165
  Q[1] = Q[1] / np.sqrt((Q[1] * Q[1]).sum(axis=1, keepdims=True))  # :167

  lb = dtype(1.0e-4)
  valid = []
  for ii in range(1, T_it + 1):
    z = A @ Q[ii]  # :173
    alpha[ii] = (Q[ii] * z).sum(axis=1, keepdims=True)  # :174
    z = z - alpha[ii] * Q[ii] - beta[ii - 1] * Q[ii - 1]  # :175
    if use_reorth and ii > 1:  # :177-189 (two passes of sequential Gram-Schmidt)
      for _ in range(2):
        for jj in range(1, ii):
          z = z - (z * Q[jj]).sum(axis=1, keepdims=True) / (
              (Q[jj] * Q[jj]).sum(axis=1, keepdims=True) + eps) * Q[jj]
    beta[ii] = np.sqrt((z * z).sum(axis=1, keepdims=True))  # :191
    ok = (beta[ii] >= lb).astype(dtype)  # :195
    valid.append(ok if ii == 1 else valid[-1] * ok)  # :196-199
    Q[ii + 1] = (z * valid[-1]) / (beta[ii] + eps)  # :202

  alpha = np.concatenate(alpha[1:], axis=1)[:, :, 0]  # B x T
  raw_betas = np.concatenate(beta[1:], axis=1)[:, :, 0]
  beta = np.concatenate(beta[1:-1], axis=1)[:, :, 0] if T_it > 1 else np.zeros((B, 0), dtype)
  valid = np.concatenate(valid, axis=1)[:, :, 0]  # B x T
  idx_mask = valid.sum(axis=1).astype(np.int64)  # :209
  if m is not None:
    idx_mask = np.minimum(idx_mask, m[:, :, 0].sum(axis=1).astype(np.int64))  # :210-211
  for b in range(B):
    if idx_mask[b] < valid.shape[1]:
      valid[b, idx_mask[b]:] = 0.0  # :213-215
  alpha = alpha * valid  # :218  (QUIRK 1)
  beta = beta * valid[:, :-1]  # :219

  T = np.zeros((B, T_it, T_it), dtype=dtype)
  for b in range(B):  # :222-226
    T[b] = np.diag(alpha[b]) + np.diag(beta[b], 1) + np.diag(beta[b], -1)
  Qm = np.concatenate(Q[1:-1], axis=2)  # B x N x T   :229
  Q_mask = np.repeat(valid[:, None, :], N, axis=1)  # :230  (QUIRK 2)
  for b in range(B):
    if idx_mask[b] < Q_mask.shape[1]:
      Q_mask[b, idx_mask[b]:, :] = 0.0  # :233-235 (QUIRK 3: zeroes NODE rows)
  Qm = Qm * Q_mask

  if T_it < K:  # :240-245
    Tp = np.zeros((B, K, K), dtype=dtype)
    Tp[:, :T_it, :T_it] = T
    Qp = np.zeros((B, N, K), dtype=dtype)
    Qp[:, :, :T_it] = Qm
    T, Qm = Tp, Qp
  if return_raw_betas:
    return T, Qm, raw_betas
  return T, Qm


def ada_graph_laplacian(node_feat, adj_mask, dtype=np.float32):
  """model/ada_lanczos_net.py:101-137.  node_feat [B,N,D], adj_mask [B,N,N] -> L [B,N,N]."""
  X = np.asarray(node_feat, dtype=dtype)
  adj = np.asarray(adj_mask, dtype=dtype)
  B, N, _ = X.shape
  # meshgrid(range(N), range(N)) -> idx_row[k] = k % N, idx_col[k] = k // N   (:115-118)
  diff = X[:, None, :, :] - X[:, :, None, :]  # [B, col, row, D]: X[row] - X[col]
  dist2 = (diff * diff).sum(axis=3).reshape(B, N * N)  # :120-121
  sigma2 = dist2.mean(axis=1, keepdims=True)  # :126
  A = np.exp(-dist2 / sigma2).reshape(B, N, N) * adj  # :128-129
  row_sum = A.sum(axis=2, keepdims=True)
  pad = (row_sum == 0.0).astype(dtype)  # :131-132
  Dm = 1.0 / np.power(row_sum + pad, dtype(0.5))  # :133-134
  return Dm * A * Dm.transpose(0, 2, 1)  # :135


# ---------------------------------------------------------------------------------------------
# Full AdaLanczosNet forward (model/ada_lanczos_net.py:250-368)
# ---------------------------------------------------------------------------------------------
ADA_HIDDEN = 4096  # model/ada_lanczos_net.py:56-62


def make_ada_params(cfg, seed, mlp_hidden=ADA_HIDDEN, bias_scale=0.05):
  """Deterministic parameters keyed like the reference AdaLanczosNet state_dict.
  input_dim is overridden to num_atom (model/ada_lanczos_net.py:40)."""
  rs = np.random.RandomState(seed)
  K, S = cfg['num_eig_vec'], len(cfg['long_diffusion_dist'])
  din = cfg['num_atom']
  dim_list = [din] + list(cfg['hidden_dim']) + [cfg['output_dim']]
  n_chan = len(cfg['short_diffusion_dist']) + S + cfg['num_bond_type'] + 1
  P = {}

  def lin(name, fan_out, fan_in):
    a = np.sqrt(6.0 / (fan_in + fan_out))
    P[name + '.weight'] = rs.uniform(-a, a, size=(fan_out, fan_in)).astype(np.float32)
    P[name + '.bias'] = (bias_scale * rs.uniform(-1, 1, size=(fan_out,))).astype(np.float32)

  for tt in range(cfg['num_layer']):
    lin('filter.%d' % tt, dim_list[tt + 1], dim_list[tt] * n_chan)
  lin('filter.%d' % cfg['num_layer'], dim_list[-1], dim_list[-2])
  P['embedding.weight'] = rs.randn(cfg['num_atom'], din).astype(np.float32)
  for tt in range(cfg['num_layer']):
    lin('spectral_filter.%d.0' % tt, mlp_hidden, K * K * S)
    lin('spectral_filter.%d.2' % tt, mlp_hidden, mlp_hidden)
    lin('spectral_filter.%d.4' % tt, mlp_hidden, mlp_hidden)
    lin('spectral_filter.%d.6' % tt, K * K * S, mlp_hidden)
  lin('att_func.0', 1, dim_list[-2])
  return P


def ada_t_powers(T, dists, dtype=np.float32):
  """T_list of model/ada_lanczos_net.py:262-270: sequential TT = TT @ T, collected at the
  requested powers; returned concatenated on dim 2 ([B, K, S*K]) like `torch.cat(T_list, dim=2)`."""
  T = np.asarray(T, dtype=dtype)
  out, TT = [], T
  for ii in range(1, max(dists) + 1):
    if ii in dists:
      out.append(TT)
    TT = TT @ T
  return np.concatenate(out, axis=2)


def ada_spectral_filter_dd(P, cfg, T, layer_idx, dtype=np.float32):
  """Symmetrised DD [B,K,K,S] of model/ada_lanczos_net.py:273-278."""
  dists = list(cfg['long_diffusion_dist'])
  B, K = T.shape[0], T.shape[1]
  S = len(dists)
  h = ada_t_powers(T, dists, dtype).reshape(B, -1)
  pre = 'spectral_filter.%d.' % layer_idx
  for ii in (0, 2, 4):
    h = np.maximum(h @ P[pre + '%d.weight' % ii].T.astype(dtype) + P[pre + '%d.bias' % ii], 0)
  h = h @ P[pre + '6.weight'].T.astype(dtype) + P[pre + '6.bias']
  DD = h.reshape(B, K, K, S)
  return (DD + DD.transpose(0, 2, 1, 3)) * dtype(0.5)


def ada_lanczos_net_forward(P, cfg, node_feat, L, mask, q1, dtype=np.float32, TQ=None,
                            return_raw_betas=False):
  """score [B,P] of AdaLanczosNet (eval mode).  q1: raw start vector [B,N] (the reference draws
  torch.randn(B,N,1) at :161).  TQ: optional precomputed (T, Q) to decouple stage tests."""
  P = {k: np.asarray(v, dtype=dtype) for k, v in P.items()}
  L = np.asarray(L, dtype=dtype)
  B, N = L.shape[0], L.shape[1]
  short = list(cfg['short_diffusion_dist'])
  S = len(cfg['long_diffusion_dist'])
  E1 = cfg['num_bond_type'] + 1
  state = P['embedding.weight'][np.asarray(node_feat)]  # :306
  if TQ is None:
    adj = (L[:, :, :, 0] != 0).astype(dtype)  # :310-311
    Le = ada_graph_laplacian(state, adj, dtype)  # :312
    T, Q, raw_betas = ada_lanczos_layer(Le, mask, q1, cfg['num_eig_vec'], True, dtype,
                                        return_raw_betas=True)  # :315 (F7: reorth on)
  else:
    raw_betas = None
    T, Q = (np.asarray(x, dtype=dtype) for x in TQ)
  for tt in range(cfg['num_layer']):
    msg = []
    DD = ada_spectral_filter_dd(P, cfg, T, tt, dtype)
    if short:  # :328-333
      tmp = state
      for ii in range(1, max(short) + 1):
        tmp = L[:, :, :, 0] @ tmp
        if ii in short:
          msg.append(tmp)
    for s in range(S):  # :280-281 + :336-338
      Ls = Q @ DD[:, :, :, s] @ Q.transpose(0, 2, 1)
      msg.append(Ls @ state)
    for e in range(E1):  # :341-342
      msg.append(L[:, :, :, e] @ state)
    msg = np.concatenate(msg, axis=2).reshape(B * N, -1)
    state = np.maximum(msg @ P['filter.%d.weight' % tt].T + P['filter.%d.bias' % tt], 0)
    state = state.reshape(B, N, -1)
  flat = state.reshape(B * N, -1)
  nl = cfg['num_layer']
  y = flat @ P['filter.%d.weight' % nl].T + P['filter.%d.bias' % nl]
  att = 1.0 / (1.0 + np.exp(-(flat @ P['att_func.0.weight'].T + P['att_func.0.bias'])))
  y = (att * y).reshape(B, N, -1)
  m = np.asarray(mask).astype(bool)
  score = np.stack([y[b, m[b], :].mean(axis=0) for b in range(B)]).astype(dtype)
  if return_raw_betas:
    return score, (T, Q), raw_betas
  return score, (T, Q)
